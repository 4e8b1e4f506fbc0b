#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/k1
python tools/bench_wgrad_lp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k1/new.txt | grep 1x1x1
MI355_WGRAD_LP_TR=0 python tools/bench_wgrad_lp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k1/old.txt | grep 1x1x1
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/k1/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/k1/pytest_gpu.log; tail -5 gpurun_out/k1/pytest_gpu.log
