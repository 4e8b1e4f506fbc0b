#!/bin/bash
# scratch per-call script (GPU box)
for a in "fp32 2" "bf16 4"; do
  echo "== new $a"; python tools/bench_upsample.py $a 2>&1 | grep -v amdgpu
  echo "== rows off $a"; MI355_UPSAMPLE_ROWS=0 python tools/bench_upsample.py $a 2>&1 | grep -v amdgpu
done
python -m pytest tests/test_ops_gpu.py tests/test_act_storage_gpu.py tests/test_model_gpu.py -q -x -k "upsample or pointwise or odd or unet3d" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c3 --no-precision-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 step', d['value'], d['ms_per_step'])"
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
