#!/bin/bash
# scratch per-call script (GPU box)
python -m pytest tests/test_act_storage_gpu.py -q -x -k "1x1x1" 2>&1 | tail -2
python tools/bench_k1_fwd.py 2>&1 | grep -v amdgpu
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
MI355_K1_STREAM=0 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 K1_STREAM=0', d['value'], d['ms_per_step'])"
python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-c3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 b2', d['value'], d['ms_per_step'])"
