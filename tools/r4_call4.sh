#!/bin/bash
# GPU call 4 of round 4: conv3d_k3_lp_zring2 -- GPU twins of its emulator tests, per-layer and whole-step A/B against the tile kernel and the
# round-3 plane-ring kernel, SQ counters and PMC traffic of single launches (32->32 and 64->32 @128^3), 16-bit audits.
out=gpurun_out/r4z2; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "zring" > $out/pytest_zring.log 2>&1; echo "pytest rc=$?" >> $out/pytest_zring.log; tail -4 $out/pytest_zring.log
for f in tile zring1 auto; do
  echo "== MI355_BF16_FORM=$f"; MI355_BF16_FORM=$f PREC=bf16 timeout 300 python tools/bench_conv_layers.py 2>&1 | tail -32
done > $out/bf16_layers_forms.txt 2>&1
paste <(grep -A31 "FORM=tile" $out/bf16_layers_forms.txt | cut -c1-50) <(grep -A31 "FORM=zring1" $out/bf16_layers_forms.txt | cut -c36-50) <(grep -A31 "FORM=auto" $out/bf16_layers_forms.txt | cut -c36-50)
for f in tile zring1 auto; do
  echo -n "bf16 step MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --precision bf16 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo -n "c3 (bf16, batch 4) MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --config c3 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $out/bf16_form_ab.txt
tools/sq_counters.sh r4z2sq "bf16 32 32 128 fwdplain" "bf16 64 32 128 fwdplain" "bf16 64 32 128 fwd" "bf16 64 64 64 fwdnormmom" > /dev/null 2>&1
cat gpurun_out/r4z2sq/sq_counters_conv_kernels.txt | grep -v "^   SQ_W\|SQ_ACTIVE" | head -60
export TMPDIR=/tmp; root=$(pwd)
for c in "bf16 32 32 128 fwdplain" "bf16 64 32 128 fwdplain"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$root/$out/pmc_$ctr; rm -rf $d
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -f csv -d $d -o sq -- python $root/tools/one_conv.py $c 3 > /dev/null 2>&1)
    echo "$c $ctr: $(python tools/pmc_kernel.py $d/sq_counter_collection.csv conv3d | tr '\n' ' ')"; rm -rf $d
  done
done | tee $out/pmc_single_launches.txt
timeout 600 python -m pytest tests/test_launch_audit.py -m gpu -q -k "16bit_train_step_gpu" > $out/pytest_audit16.log 2>&1; tail -3 $out/pytest_audit16.log
