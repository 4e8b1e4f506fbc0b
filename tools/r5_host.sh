#!/bin/bash
# Round 5, host side of a step on the GPU box: bench.py's `per_rank_host_enqueue_ms_per_step` with and without the caches (MI355_HOST_CACHES=0) and the cProfile breakdown
# on an idle device; then the GPU suite (every test goes through the cached parameter list / gradient views / descriptors) without the two
# 1-minute sliding-window cases.  Usage: tools/r5_host.sh <tag>
tag=${1:-r5h}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-c3 --no-precision-modes"
timeout 300 python bench.py $B > $out/bench_fp32.json 2> $out/bench_fp32.err
MI355_HOST_CACHES=0 timeout 300 python bench.py $B > $out/bench_fp32_nocache.json 2>> $out/bench_fp32.err
timeout 300 python bench.py $B --precision bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
MI355_HOST_CACHES=0 timeout 300 python bench.py $B --precision bf16 > $out/bench_bf16_nocache.json 2>> $out/bench_bf16.err
for f in fp32 fp32_nocache bf16 bf16_nocache; do python - "$out/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:14s} ms_per_step {j['ms_per_step']:.2f}  host_enqueue {j['per_rank_host_enqueue_ms_per_step']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done | tee $out/summary.txt
timeout 300 python tools/host_enqueue.py --gpu --steps 10 --profile > $out/profile_fp32.txt 2>&1; head -3 $out/profile_fp32.txt
MI355_HOST_CACHES=0 timeout 300 python tools/host_enqueue.py --gpu --steps 10 > $out/idle_fp32_nocache.txt 2>&1; head -1 $out/idle_fp32_nocache.txt
timeout 620 python -m pytest tests -m gpu -q -x --durations=5 -k "not c5_sliding_window" > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -8 $out/pytest_gpu.log
