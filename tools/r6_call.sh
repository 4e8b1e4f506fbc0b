#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/wlp
python -m pytest tests/test_act_storage_gpu.py -q -x -k "weight_gradients or fp16" > gpurun_out/wlp/pytest.txt 2>&1
tail -3 gpurun_out/wlp/pytest.txt
python tools/bench_wgrad_lp.py > gpurun_out/wlp/new.txt 2>&1
MI355_WGRAD_LP_TR=0 python tools/bench_wgrad_lp.py > gpurun_out/wlp/old.txt 2>&1
cat gpurun_out/wlp/new.txt gpurun_out/wlp/old.txt
