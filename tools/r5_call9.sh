#!/bin/bash
# Round 5, GPU call 9: same-box comparison of the d8 forms: tree (input requested 4 phases ahead, activation at the bottom of the phase),
# libvar_d8v1 (commit cdf981a: plain forms only, input 3 phases ahead), libvar_w8only.
out=gpurun_out/r5i; mkdir -p $out
export TMPDIR=/tmp
MODES=plain+gnb,norm+moments timeout 500 python tools/bench_conv_layers.py tree tools/libvar_d8v1.so tools/libvar_w8only.so > $out/conv_layers.txt 2>&1; tail -23 $out/conv_layers.txt
