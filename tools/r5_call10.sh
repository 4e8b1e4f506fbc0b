#!/bin/bash
# Round 5, GPU call 10: what is left in conv3d_wino2d_d8? Ablations (wrong results, timing only): 1 no input DMA, 2 no weight DMA, 4 weight
# fragments from constants instead of the LDS slab.
out=gpurun_out/r5j; mkdir -p $out
export TMPDIR=/tmp
MODES=plain,norm timeout 500 python tools/bench_conv_layers.py tree tools/libvar_d8abl1.so tools/libvar_d8abl2.so tools/libvar_d8abl3.so tools/libvar_d8abl4.so tools/libvar_d8abl7.so tools/libvar_w8only.so > $out/conv_layers.txt 2>&1; tail -23 $out/conv_layers.txt
