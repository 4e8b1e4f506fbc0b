#!/bin/bash
out=gpurun_out/r6k
mkdir -p $out
python -m pytest tests/test_act_storage_gpu.py tests/test_ops_gpu.py tests/test_launch_audit.py -m gpu -q -x -k "first_layer or 16bit_train_step_gpu or c3" 2>&1 | tail -4
for f in 0 1; do
  for cfg in "--config c3" "--precision bf16"; do
    MI355_C4_BWD=$f python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4_bwd=$f $cfg', d['ms_per_step'], d['value'])"
  done
done
