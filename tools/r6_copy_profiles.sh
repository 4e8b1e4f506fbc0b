#!/bin/bash
# After `gpurun -- 'bash tools/r6_final.sh'`: copy what that call wrote under gpurun_out/ to the tracked names profiles/README.md lists.
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/r6f/bench.json profiles/r6_bench_fp32.json
cp gpurun_out/r6f/trace/bench_kernel_stats.csv profiles/r6_bench_fp32_kernel_stats.csv
cp gpurun_out/r6f/hbm_traffic_pmc.csv profiles/r6_bench_fp32_hbm_traffic_pmc.csv
cp gpurun_out/r6f/bench_bf16.json profiles/r6_bf16_bench.json
cp gpurun_out/r6fbf16/trace_bf16/bench_kernel_stats.csv profiles/r6_bf16_kernel_stats.csv
cp gpurun_out/r6fbf16/hbm_traffic_pmc_bf16.csv profiles/r6_bf16_hbm_traffic_pmc.csv
cp gpurun_out/r6f/bench_c3.json profiles/r6_c3_bench.json
cp gpurun_out/r6fc3/trace_bf16/bench_kernel_stats.csv profiles/r6_c3_kernel_stats.csv
cp gpurun_out/r6fc3/hbm_traffic_pmc_bf16.csv profiles/r6_c3_hbm_traffic_pmc.csv
cp gpurun_out/r6f/other_configs.txt profiles/r6_other_configs.txt
{ cat gpurun_out/r6f/smoke.log | tail -1; grep -v "amdgpu.ids" gpurun_out/r6f/pytest_gpu.log | tail -60; } > profiles/r6_pytest_gpu_tail.log
python - <<'PY'
import json
for f in ("profiles/r6_bench_fp32.json", "profiles/r6_bf16_bench.json", "profiles/r6_c3_bench.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["value"], d["unit"], d["ms_per_step"], "ms | roofline", r["kernel"], r["frac"], "traffic", r["traffic"], "clock", r.get("shader_clock_ghz"),
          "| first_layer", d.get("first_layer", {}).get("hbm_frac"), "| c3", (d.get("c3") or {}).get("volumes_per_s_per_gpu"))
PY
