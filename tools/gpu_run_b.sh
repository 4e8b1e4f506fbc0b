#!/bin/bash
out=$(pwd)/gpurun_out/r1i; mkdir -p $out; root=$(pwd)
cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for cfg in "fp32 32 32 128 fwd" "fp32 128 128 64 fwd" "fp32 256 256 32 fwd" "fp32 128 128 64 wgrad"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $A -f csv -d $out/$tag.a -o p -- python $root/tools/one_conv.py $cfg > /dev/null 2>&1
  echo "=== $cfg"; python $root/tools/pmc_kernel.py $out/$tag.a/p_counter_collection.csv conv3d
done
