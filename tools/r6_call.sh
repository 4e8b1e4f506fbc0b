#!/bin/bash
out=gpurun_out/r6g
mkdir -p $out
for f in 0 1 0 1; do
  MI355_C4_BWD=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes --no-c3 > $out/bench_c4bwd$f.json 2> $out/bench_c4bwd$f.err
  python - <<PY
import json
d=json.load(open("$out/bench_c4bwd$f.json"))
fl=d["first_layer"]
print("c4_bwd=$f", d["ms_per_step"], d["value"], d["roofline"]["frac"], fl["kernels_ms_per_launch"], fl["hbm_frac"])
PY
done
python -m pytest tests/test_launch_audit.py tests/test_headline_parity_gpu.py -m gpu -q -x -k "headline" 2>&1 | tail -3
