#!/bin/bash
# Round 5, GPU call 4: conv3d_wino2d_r8 after the register clean-up (straight-line staging, scalar-base loads) against conv3d_wino2d_w8.
out=gpurun_out/r5d; mkdir -p $out
export TMPDIR=/tmp
MODES=plain,norm+moments timeout 400 python tools/bench_conv_layers.py tree tools/libvar_w8.so > $out/conv_layers.txt 2>&1; tail -23 $out/conv_layers.txt
tools/sq_counters.sh r5d "fp32 32 32 128 fwdplain"
