#!/bin/bash
# Runs on the GPU box: the Winograd weight gradient of the in-tree library against a variant build, per layer.
#   tools/wgrad_lib_ab.sh <tag> <variant.so>
tag=${1:-wgrad_lib_ab}; var=$2; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_wino_gpu.py -q -m gpu -k "wgrad_wino_ring" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
for c in "32 32 128" "64 32 128" "64 64 64" "128 64 64" "128 128 32" "256 256 32"; do
  for l in tree $var; do
    echo -n "wgrad $c  $l: "
    if [ $l = tree ]; then python tools/one_conv.py fp32 $c wgrad 12 2>/dev/null | tail -1; else ONE_CONV_LIB=$l python tools/one_conv.py fp32 $c wgrad 12 2>/dev/null | tail -1; fi
  done
done | tee $out/wgrad.txt
