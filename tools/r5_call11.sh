#!/bin/bash
# Round 5, GPU call 11: conv3d_wino2d_d8 with the DMA requests between the MFMA pairs and the first-use weight fragments read one phase
# ahead, against conv3d_wino2d_w8 (libvar_w8only).
out=gpurun_out/r5k; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wino_gpu.py -q > $out/test_wino_gpu.txt 2>&1; tail -3 $out/test_wino_gpu.txt
MODES=plain+gnb,norm+moments,plain timeout 400 python tools/bench_conv_layers.py tree tools/libvar_w8only.so > $out/conv_layers.txt 2>&1; tail -33 $out/conv_layers.txt
timeout 600 python bench.py --no-cpu-baseline --no-precision-modes > $out/bench.json 2> $out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5k/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('final_loss'))
PY
