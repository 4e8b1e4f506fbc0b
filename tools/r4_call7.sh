#!/bin/bash
# GPU call 7: -amdgpu-mfma-vgpr-form on the plane-ring kernels (tree) against the same sources without it (tools/libvar_noflag.so), per layer
# and whole step; GPU twins of the zring tests on the flagged build; the graphed step with its side stream captured.
out=gpurun_out/r4vf; mkdir -p $out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_graph.py -m gpu -q -x -k "zring or graph" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for f in auto zring1; do
  echo "== MI355_BF16_FORM=$f   (columns: no flag | -amdgpu-mfma-vgpr-form)"; MI355_BF16_FORM=$f PREC=bf16 timeout 400 python tools/bench_conv_layers.py tools/libvar_noflag.so tree 2>&1 | tail -32
done | tee $out/bf16_layers_vgpr_form.txt
for f in zring1 auto; do
  echo -n "bf16 step MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --precision bf16 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo -n "c3 MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --config c3 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $out/bf16_form_ab.txt
for g in "" "--graph"; do
  for s in 1 0; do
    echo -n "fp32 step $g MI355_GRAPH_SIDE_STREAM=$s: "; MI355_GRAPH_SIDE_STREAM=$s python bench.py $g --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['step_form'])"
  done
done | tee $out/graph_ab.txt
tools/sq_counters.sh r4vfsq "bf16 32 32 128 fwdplain" "bf16 64 32 128 fwdplain" "bf16 64 64 64 fwdnormmom" > /dev/null 2>&1
grep -v "SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_VMEM" gpurun_out/r4vfsq/sq_counters_conv_kernels.txt | head -60
