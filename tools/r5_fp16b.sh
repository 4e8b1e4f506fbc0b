#!/bin/bash
out=gpurun_out/r5s; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_act_storage_gpu.py "tests/test_launch_audit.py::test_audit_16bit_train_step_gpu" tests/test_oracle_pinned.py -m gpu -q -k "fp16 or autocast" > $out/pytest_fp16.log 2>&1; tail -4 $out/pytest_fp16.log
timeout 300 python bench.py --precision fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes > $out/fp16_stored.json 2>$out/err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5s/fp16_stored.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['activation_storage'][:8])
for i in d['roofline'].get('instantiations', []): print('  ', i['kernel'], i['launches_per_step'], i['avg_launch_ms'])
PY
