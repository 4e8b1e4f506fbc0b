#!/bin/bash
# scratch per-call script (GPU box)
out=gpurun_out/r6x; mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu --durations=12 ) > $out/pytest_gpu_workers.log 2>&1; echo "rc=$?" >> $out/pytest_gpu_workers.log
tail -22 $out/pytest_gpu_workers.log
