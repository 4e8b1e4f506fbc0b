#!/bin/bash
# GPU call 3 of round 4: A/B of (a) the conflict-free A-fragment layout of conv3d_wino2d_w8 (tree vs git HEAD, per layer), (b) lagged
# side-stream weight gradients (whole step), (c) the side stream in the bf16 / C3 step; the fp16 audit with its loss scale.
out=gpurun_out/r4ab; mkdir -p $out
timeout 400 python tools/bench_conv_layers.py tools/libvar_head.so tree > $out/wino_layout_layers.txt 2>&1; cat $out/wino_layout_layers.txt | tail -34
for lag in 0 1 0 1; do
  echo -n "fp32 step MI355_WGRAD_LAG=$lag: "; MI355_WGRAD_LAG=$lag python bench.py --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $out/wgrad_lag_ab.txt
for cfg in "MI355_SIDE_STREAM=0" "MI355_WGRAD_LAG=0" "MI355_WGRAD_LAG=1"; do
  echo -n "c3 $cfg: "; env $cfg python bench.py --config c3 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo -n "bf16 batch 2 $cfg: "; env $cfg python bench.py --precision bf16 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $out/c3_side_stream_ab.txt
timeout 300 python -m pytest tests/test_launch_audit.py -m gpu -q -k "16bit_train_step_gpu" > $out/pytest_fp16.log 2>&1; tail -3 $out/pytest_fp16.log
