#!/bin/bash
out=gpurun_out/r6e
mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "first_layer_fused" 2>&1 | tail -3 > $out/pytest_c4.txt; cat $out/pytest_c4.txt
python tools/bench_first_layer.py 2>&1 | grep -v "^bf16" > $out/first_layer.txt; cat $out/first_layer.txt | cut -c1-330
