#!/bin/bash
# Round 5, GPU call 6: conv3d_wino2d_d8 (plain-input launches: LDS-DMA staging two phases ahead, register-generated A fragments) against
# conv3d_wino2d_w8 (tools/libvar_w8only.so = the tree built with -DWINO_D8=0): op tests, per-layer launch times.
out=gpurun_out/r5f; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wino_gpu.py -q > $out/test_wino_gpu.txt 2>&1; tail -5 $out/test_wino_gpu.txt
MODES=plain,plain+gnb,moments timeout 400 python tools/bench_conv_layers.py tree tools/libvar_w8only.so > $out/conv_layers.txt 2>&1; tail -33 $out/conv_layers.txt
