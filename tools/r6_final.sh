#!/bin/bash
# Final collection of round 6 (one GPU call, ~22 min of box time): smoke + the FULL GPU suite on the final kernel sources, then the
# serialized kernel traces + PMC traffic of the fp32 headline step, the bf16 step and C3, the bench lines quoting them (the PMC
# summaries are copied to where bench.py looks for them, bench.PMC_FILES, BEFORE the lines run), the other configurations.
# Afterwards, here: copy gpurun_out/r6f*/... into profiles/ as profiles/README.md lists.
out=gpurun_out/r6f; mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=30 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
tools/gpu_profile.sh r6f trace pmc
cp gpurun_out/r6f/hbm_traffic_pmc.csv profiles/r6_bench_fp32_hbm_traffic_pmc.csv
tools/trace_bf16.sh r6fbf16 > $out/trace_bf16.log 2>&1
cp gpurun_out/r6fbf16/hbm_traffic_pmc_bf16.csv profiles/r6_bf16_hbm_traffic_pmc.csv
tools/trace_bf16.sh r6fc3 --config c3 > $out/trace_c3.log 2>&1
cp gpurun_out/r6fc3/hbm_traffic_pmc_bf16.csv profiles/r6_c3_hbm_traffic_pmc.csv
tools/gpu_profile.sh r6f bench
timeout 600 python bench.py --config c3 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 300 $out/bench_c3.json
timeout 600 python bench.py --precision bf16 --no-cpu-baseline > $out/bench_bf16.json 2> $out/bench_bf16.err
{
for a in "--config c4" "--config c5" "--model dynunet" "--graph" "--precision fp16" "--precision fp16 --storage fp32" "--size 64 --batch 1"; do
  echo -n "bench.py $a: "; timeout 600 python bench.py $a --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], '|', d['value'], d['unit'], '|', d['ms_per_step'], 'ms/step |', d.get('step_form', ''), '|', d['config'].get('activation_storage', '')[:12])"
done
} | tee $out/other_configs.txt
python tools/bench_first_layer.py > $out/first_layer.txt 2>&1; tail -3 $out/first_layer.txt | cut -c1-300
