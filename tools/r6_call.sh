#!/bin/bash
# scratch per-call script (GPU box)
for i in 1 2; do
for m in 0 -1; do
MI355_MAIN_PRIORITY=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes --no-c3 --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 main priority $m', d['value'], d['ms_per_step'])"
done; done
for m in 0 -1; do
MI355_MAIN_PRIORITY=$m python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 main priority $m', d['value'], d['ms_per_step'])"
done
