#!/bin/bash
out=gpurun_out/r6j
mkdir -p $out
LAYERS=0,1,2,3,4,5 python tools/bench_conv_layers.py tools/libvar_prio0.so@2d tools/libvar_prio1.so@2d > $out/prio.txt 2>&1
cat $out/prio.txt
