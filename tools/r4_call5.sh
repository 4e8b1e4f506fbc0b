#!/bin/bash
out=gpurun_out/r4z2b; mkdir -p $out
for v in 1 2 3; do
  echo "== variant dbg$v"; ONE_CONV_LIB=tools/libvar_dbg$v.so MI355_BF16_FORM=zring timeout 120 python tools/r4_dbg.py 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $out/dbg_variants.txt
