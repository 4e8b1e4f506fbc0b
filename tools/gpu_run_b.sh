#!/bin/bash
out=$(pwd)/gpurun_out/r1g; mkdir -p $out; root=$(pwd)
cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
for cfg in "fp32 32 32 128 fwd" "fp32 64 64 64 fwd" "bf16x3 32 32 128 fwd" "bf16x3 128 128 64 fwd" "fp32 32 32 128 wgrad" "bf16x3 32 32 128 wgrad"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $A -f csv -d $out/$tag.a -o p -- python $root/tools/one_conv.py $cfg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $B -f csv -d $out/$tag.b -o p -- python $root/tools/one_conv.py $cfg > /dev/null 2>&1
  echo "=== $cfg"; python $root/tools/pmc_kernel.py $out/$tag.a/p_counter_collection.csv conv3d; python $root/tools/pmc_kernel.py $out/$tag.b/p_counter_collection.csv conv3d | grep -v avg_us
done
