#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/c3
python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c3/bench_new.json 2> gpurun_out/c3/bench_new.err
MI355_WGRAD_LP_TR=0 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c3/bench_old.json 2> gpurun_out/c3/bench_old.err
python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-c3 > gpurun_out/c3/bench_bf16_new.json 2> gpurun_out/c3/bench_bf16_new.err
(cd /tmp && export TMPDIR=/tmp && MI355_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/c3/trace -o c3 -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c3/trace.log 2>&1)
find gpurun_out/c3/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/c3/c3_kernel_stats.csv \;
rm -rf gpurun_out/c3/trace
for f in gpurun_out/c3/bench_new.json gpurun_out/c3/bench_old.json gpurun_out/c3/bench_bf16_new.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["unit"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
