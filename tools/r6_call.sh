#!/bin/bash
out=gpurun_out/r6h
mkdir -p $out
python tools/bench_stride2.py > $out/s2_base.txt 2>&1; grep -v amdgpu $out/s2_base.txt
MI355_S2_KC16=1 python tools/bench_stride2.py > $out/s2_kc16.txt 2>&1; grep -v amdgpu $out/s2_kc16.txt
