#!/bin/bash
# Runs on the GPU box: the plane-ring Winograd weight gradient against conv3d_wgrad_ring, per layer and in the whole step.
tag=${1:-wgrad_ab}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_wino_gpu.py -q -m gpu -k "wgrad_wino_ring" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
for c in "32 32 128" "64 32 128" "64 64 64" "128 128 32"; do
  for f in direct wino; do
    echo -n "wgrad $c  MI355_WGRAD_FORM=$f: "
    MI355_WGRAD_FORM=$f python tools/one_conv.py fp32 $c wgrad 12 2>/dev/null | tail -1
  done
done | tee $out/wgrad.txt
for f in direct wino; do
  echo -n "step MI355_WGRAD_FORM=$f: "
  MI355_WGRAD_FORM=$f python bench.py --no-cpu-baseline --no-precision-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], 'ms/step', d['value'], 'vol/s | dominant', r['kernel'], r['avg_launch_ms'], 'ms', r['achieved'], 'TF/s; ', {k:(round(v['s']*1e3/3,2), v['launches']//3) for k,v in r['all_kernels'].items()})"
done | tee $out/step.txt
