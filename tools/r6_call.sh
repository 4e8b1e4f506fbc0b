#!/bin/bash
# scratch per-call script (GPU box)
for i in 1 2; do
for m in 0 2 6 7 4; do
MI355_K1_STREAM_F32=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes --no-c3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 mask $m', d['value'], d['ms_per_step'])"
done; done
