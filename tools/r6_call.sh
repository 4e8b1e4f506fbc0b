#!/bin/bash
# scratch per-call script (GPU box)
python -m pytest tests/test_act_storage_gpu.py -q -x -k "3x3x3" 2>&1 | tail -2
export LAYERS="256,256,16;512,256,16"
echo "== tree"; python tools/bench_lp_tile.py 2>&1 | grep -v amdgpu
echo "== MI355_BF16_WIDE=0"; MI355_BF16_WIDE=0 python tools/bench_lp_tile.py 2>&1 | grep -v amdgpu
unset LAYERS
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
MI355_BF16_WIDE=0 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 WIDE=0', d['value'], d['ms_per_step'])"
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
python -m pytest tests/test_launch_audit.py -q -x -k "c3" 2>&1 | tail -2
