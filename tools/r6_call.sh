#!/bin/bash
# scratch per-call script (GPU box)
python -m pytest tests/test_ops_gpu.py tests/test_act_storage_gpu.py -q -x -k "dgrad or zero_insert or stride" 2>&1 | tail -2
python tools/bench_stride2.py 2>&1 | grep -v amdgpu | head -1
MI355_S2_KERNEL=0 python tools/bench_stride2.py 2>&1 | grep -v amdgpu | head -1
