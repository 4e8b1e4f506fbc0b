#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline > $out/c3_5_2.json 2>$out/c3.err
timeout 300 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline > $out/c3_20_5.json 2>>$out/c3.err
timeout 300 python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes > $out/bf16_20_5.json 2>>$out/c3.err
python - <<'PY'
import json
for f in ('c3_5_2','c3_20_5','bf16_20_5'):
    d=json.loads(open('gpurun_out/r5p/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
