#!/bin/bash
# Round 5: fp16 activation storage on the GPU: op-level twins, the per-launch audit of the 16-bit steps, the reference-graph test, the step.
out=gpurun_out/r5q; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_act_storage_gpu.py tests/test_oracle_pinned.py "tests/test_launch_audit.py::test_audit_16bit_train_step_gpu" -m gpu -q --durations=5 > $out/pytest_fp16.log 2>&1; tail -8 $out/pytest_fp16.log
timeout 300 python bench.py --precision fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes > $out/fp16_stored.json 2>$out/err.log
timeout 300 python bench.py --precision fp16 --storage fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes > $out/fp16_fp32tensors.json 2>>$out/err.log
python - <<'PY'
import json
for f in ('fp16_stored','fp16_fp32tensors'):
    d=json.loads(open('gpurun_out/r5q/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['config']['activation_storage'][:20], d['final_loss'])
PY
