#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/s2w
python tools/bench_stride2.py > gpurun_out/s2w/new.txt 2>&1
MI355_S2_KERNEL=0 python tools/bench_stride2.py > gpurun_out/s2w/old.txt 2>&1
python -m pytest tests/test_ops_gpu.py tests/test_act_storage_gpu.py -q -x -k "wgrad or stride" > gpurun_out/s2w/pytest.txt 2>&1
tail -3 gpurun_out/s2w/pytest.txt; cat gpurun_out/s2w/new.txt gpurun_out/s2w/old.txt
