#!/bin/bash
out=gpurun_out/r5r; mkdir -p $out
export TMPDIR=/tmp
MODES=plain+gnb,norm+moments timeout 500 python tools/bench_conv_layers.py tree tools/libvar_zb4.so tools/libvar_zb16.so tools/libvar_zb64.so > $out/conv_layers.txt 2>&1; tail -23 $out/conv_layers.txt
