#!/bin/bash
out=gpurun_out/r6d
mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "first_layer_fused" 2>&1 | tail -4 > $out/pytest_c4.txt; cat $out/pytest_c4.txt
python tools/bench_first_layer.py > $out/first_layer.txt 2>&1; tail -4 $out/first_layer.txt
for f in 0 1; do
  MI355_C4_BWD=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes --no-c3 > $out/bench_c4bwd$f.json 2> $out/bench_c4bwd$f.err
  python - <<PY
import json
d=json.load(open("$out/bench_c4bwd$f.json"))
print("c4_bwd=$f", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["first_layer"])
PY
done
python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -45 > $out/pytest_gpu.txt
tail -32 $out/pytest_gpu.txt
