#!/bin/bash
# scratch per-call script (GPU box)
tools/sq_counters.sh r6sqw "bf16 128 128 64 fwdplain" "bf16 128 128 64 fwdnormmom" "bf16 256 256 32 fwdplain"
MI355_BF16_WIDE=0 tools/sq_counters.sh r6sq0 "bf16 128 128 64 fwdplain" "bf16 128 128 64 fwdnormmom" "bf16 256 256 32 fwdplain"
echo "== wide"; cat gpurun_out/r6sqw/sq_counters_conv_kernels.txt
echo "== MI355_BF16_WIDE=0"; cat gpurun_out/r6sq0/sq_counters_conv_kernels.txt
