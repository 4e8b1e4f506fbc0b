#!/bin/bash
# Final collection of round 4 (≈11 min of box time): serialized kernel traces + PMC traffic of the fp32 headline step, the bf16 step and C3;
# the bench lines quoting them; the other configurations. The PMC summaries are copied to where bench.py looks for them (profiles/,
# bench.PMC_FILES) BEFORE the bench lines run, so that `roofline.traffic` of each line is the traffic of this very tree.
tools/gpu_profile.sh r4f trace pmc
cp gpurun_out/r4f/hbm_traffic_pmc.csv profiles/r4_bench_fp32_hbm_traffic_pmc.csv
tools/gpu_profile.sh r4f bench
tools/trace_bf16.sh r4fbf16 > gpurun_out/r4f/trace_bf16.log 2>&1
cp gpurun_out/r4fbf16/hbm_traffic_pmc_bf16.csv profiles/r4_bf16_hbm_traffic_pmc.csv
tools/trace_bf16.sh r4fc3 --config c3 > gpurun_out/r4f/trace_c3.log 2>&1
cp gpurun_out/r4fc3/hbm_traffic_pmc_bf16.csv profiles/r4_c3_hbm_traffic_pmc.csv
out=gpurun_out/r4f
timeout 600 python bench.py --config c3 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 400 $out/bench_c3.json
timeout 600 python bench.py --precision bf16 --no-cpu-baseline > $out/bench_bf16.json 2> $out/bench_bf16.err
{
for a in "--config c4" "--config c5" "--model dynunet" "--graph" "--precision fp16" "--size 64 --batch 1"; do
  echo -n "bench.py $a: "; timeout 600 python bench.py $a --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], '|', d['value'], d['unit'], '|', d['ms_per_step'], 'ms/step |', d.get('step_form', ''))"
done
} | tee $out/other_configs.txt
