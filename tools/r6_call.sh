#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/k1
export LAYERS="32,32,32"
python -m pytest tests/test_ops_gpu.py tests/test_act_storage_gpu.py -q -x -k "conv_wgrad or weight_gradients" 2>&1 | tail -2
K1_FP32=1 python tools/bench_wgrad_lp.py 2 2>&1 | grep 1x1x1 | tee gpurun_out/k1/fp32_new.txt
K1_FP32=1 MI355_WGRAD_K1_STREAM=0 python tools/bench_wgrad_lp.py 2 2>&1 | grep 1x1x1 | tee gpurun_out/k1/fp32_old.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c3 --no-precision-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 step', d['value'], d['ms_per_step'])"
MI355_WGRAD_K1_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c3 --no-precision-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 step, old 1x1x1 wgrad', d['value'], d['ms_per_step'])"
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
