#!/bin/bash
out=gpurun_out/r1e; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -8 $out/pytest_gpu.log
for p in fp32 bf16x6 bf16x3 bf16; do
  timeout 600 python bench.py --precision $p --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err
  python -c "
import json,sys
d=json.load(open('$out/bench_$p.json')); r=d['roofline']
print('$p', d['value'], 'vol/s', d['ms_per_step'], 'ms loss', d['final_loss'], '|', r['kernel'], r['achieved'], 'TF/s share', r['share_of_step'])
for k,v in list(r['all_kernels'].items())[:6]: print('    ', k, v)
"
done
timeout 600 python bench.py --model dynunet --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_dyn_fp32.json 2> $out/bench_dyn.err; cut -c1-400 $out/bench_dyn_fp32.json
timeout 600 python bench.py --model dynunet --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --precision bf16x3 > $out/bench_dyn_bf16x3.json 2>> $out/bench_dyn.err; cut -c1-400 $out/bench_dyn_bf16x3.json
