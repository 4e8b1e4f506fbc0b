#!/bin/bash
# Runs on the GPU box: SQ counters (MFMA-pipe busy cycles, LDS waits / bank conflicts, instruction mix) of single launches of the dominant
# fp32 kernels, each counter group in its own rocprofv3 --pmc pass (kernel-trace only, as the pool requires).
# Usage: tools/sq_counters.sh <tag> ["<precision> <cin> <cout> <size> <fwd|fwdplain|wgrad>" ...]   -> gpurun_out/<tag>/sq_counters_conv_kernels.txt
tag=${1:-r2}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
res=$root/$out/sq_counters_conv_kernels.txt
: > $res
groups=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE")
cases=("fp32 32 32 128 fwd" "fp32 64 64 64 fwd" "fp32 32 32 128 fwdplain" "fp32 32 32 128 wgrad" "fp32 128 128 32 wgrad")
if [ $# -gt 0 ]; then cases=("$@"); fi
for c in "${cases[@]}"; do
  echo "=== $c" >> $res
  i=0
  for g in "${groups[@]}"; do
    d=$root/$out/sq_$i; rm -rf $d
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $g -f csv -d $d -o sq -- python $root/tools/one_conv.py $c 3 > /dev/null 2>&1)
    python tools/pmc_kernel.py $d/sq_counter_collection.csv conv3d >> $res 2>/dev/null
    rm -rf $d
    i=$((i+1))
  done
done
wc -l $res
