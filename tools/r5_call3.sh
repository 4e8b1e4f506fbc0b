#!/bin/bash
# Round 5, GPU call 3: conv3d_wino2d_r8 (A fragments generated in registers) against conv3d_wino2d_w8 (tools/libvar_w8.so = the tree
# built with -DWINO_R8=0): GPU twins of the Winograd op tests, the per-layer launch times of both on this box, SQ counters of r8.
out=gpurun_out/r5c; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q > $out/test_wino_gpu.txt 2>&1; tail -3 $out/test_wino_gpu.txt
timeout 400 python tools/bench_conv_layers.py tree tools/libvar_w8.so > $out/conv_layers.txt 2>&1; tail -33 $out/conv_layers.txt
tools/sq_counters.sh r5c "fp32 32 32 128 fwdnormmom" "fp32 256 256 32 fwdnormmom"
