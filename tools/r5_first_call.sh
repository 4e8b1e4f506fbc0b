#!/bin/bash
# First GPU call of round 5 (≈4 min of box time + the box's first `import torch`): the two measurements round 4 ended on needing.
#  1. tools/micro/tr_read_probe.hip: what ds_read_b64_tr_b16 hands each lane (NEXT.md candidate 1: an LDS-transposing 16-bit weight gradient)
#  2. SQ counters of the 16-bit weight gradient (never collected: is it MFMA-, LDS-, VALU- or barrier-bound?) with fp32 tensors
#     (tools/one_conv.py makes fp32 activations) on the 128^3 and the 64-channel shapes
out=gpurun_out/r5a; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe 2> $out/tr_read_build.log && tools/micro/tr_read_probe > $out/tr_read_probe.txt 2>&1
head -40 $out/tr_read_probe.txt
tools/sq_counters.sh r5a "bf16 32 32 128 wgrad" "bf16 64 64 64 wgrad" "bf16 128 128 32 wgrad"
cat gpurun_out/r5a/sq_counters_conv_kernels.txt | head -60
