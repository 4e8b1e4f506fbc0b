#!/bin/bash
# Round 5, GPU call 12: what do the DMA requests of conv3d_wino2d_d8 cost? (timing only) 32: the loop never waits for them; 64: every input lane
# fetches the same 16 bytes; 128: every weight lane does; 3: no requests at all.
out=gpurun_out/r5l; mkdir -p $out
export TMPDIR=/tmp
MODES=plain timeout 500 python tools/bench_conv_layers.py tree tools/libvar_d8x32.so tools/libvar_d8x64.so tools/libvar_d8x128.so tools/libvar_d8x192.so tools/libvar_d8a3.so tools/libvar_w8only.so > $out/conv_layers.txt 2>&1; tail -13 $out/conv_layers.txt
