#!/bin/bash
out=gpurun_out/r6m
mkdir -p $out
python -m pytest tests/test_ops_gpu.py tests/test_act_storage_gpu.py -m gpu -q -x -k "stride or s2 or moments or conv_fwd" 2>&1 | tail -3
MI355_S2_KERNEL=0 python tools/bench_stride2.py 2>&1 | grep "32->32"
python tools/bench_stride2.py 2>&1 | grep "32->32"
