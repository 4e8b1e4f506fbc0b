#!/bin/bash
# scratch per-call script (GPU box)
python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests/test_act_storage_gpu.py tests/test_wino_gpu.py -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['c3']['volumes_per_s_per_gpu'])"
