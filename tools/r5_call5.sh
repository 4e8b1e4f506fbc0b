#!/bin/bash
# Round 5, GPU call 5: which latency chain bounds the Winograd forward kernel? Ablations (results wrong by construction, timing only) of
# conv3d_wino2d_r8 (commit 909edce: no transform through LDS) and conv3d_wino2d_w8: 1 = no input loads, 2 = no weight loads, 4 = no transform.
out=gpurun_out/r5e; mkdir -p $out
export TMPDIR=/tmp
MODES=plain timeout 600 python tools/bench_conv_layers.py tree tools/libvar_w8abl3.so tools/libvar_w8abl7.so tools/libvar_r8.so tools/libvar_r8abl1.so tools/libvar_r8abl2.so tools/libvar_r8abl3.so > $out/conv_layers.txt 2>&1; tail -13 $out/conv_layers.txt
