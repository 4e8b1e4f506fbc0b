#!/bin/bash
# Runs on the GPU box: 16-bit activation storage -- op-level GPU tests (tile and plane-ring forms), the audited steps (64^3 in three modes,
# C3 at 128^3 x 4), the plane-ring tests of the fp32-tensor path (unchanged kernels, retyped sources), and the bf16 / C3 step with either storage.
out=gpurun_out/r4st; mkdir -p $out
timeout 1200 python -m pytest tests/test_act_storage_gpu.py tests/test_launch_audit.py tests/test_ops_gpu.py -m gpu -q -x -k "act_storage or 16bit_train_step_gpu or c3_bf16 or zring or bf16_paths" -s > $out/tests.txt 2>&1
echo "pytest rc=$?" >> $out/tests.txt; grep -E "passed|failed|rc=|c3 logits|prologue forms" $out/tests.txt | tail -6
step() { python bench.py --no-cpu-baseline --no-precision-modes "$@" 2> $out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
ak=r['all_kernels']; n=r['launches']//r['launches_per_step'] if r.get('launches_per_step') else 3
print(d['ms_per_step'], 'ms/step', d['value'], 'vol/s | frac', r['frac'], 'hbm_frac', r.get('hbm_frac'), '|', {k[:28]: round(v['s']*1e3/n, 2) for k, v in ak.items()})" || tail -5 $out/err.txt; }
(echo -n "bf16, bf16 storage: "; step --precision bf16
 echo -n "bf16, fp32 tensors: "; step --precision bf16 --storage fp32
 echo -n "C3, bf16 storage: "; step --config c3
 echo -n "C3, fp32 tensors: "; step --config c3 --storage fp32) | tee $out/steps.txt
