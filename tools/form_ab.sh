#!/bin/bash
# Runs on the GPU box: forms of the Winograd forward / dgrad kernel (MI355_WINO_FORM) against each other, per layer and whole step.
#   tools/form_ab.sh <tag> "<form> <form> ..." [pytest -k expression]
tag=${1:-form_ab}; forms=${2:-"tile w8"}; kexpr=${3:-"eight"}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_wino_gpu.py -q -m gpu -k "$kexpr" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
for f in $forms; do MI355_WINO_FORM=$f python tools/bench_conv_layers.py > $out/layers_$f.txt 2>&1; done
FORMS="$forms" python - <<PY | tee $out/layers.txt
import os
forms = os.environ["FORMS"].split()
rows = {}
for f in forms:
    for l in open("$out/layers_%s.txt" % f):
        if " k3 " in l or l.startswith("sum"):
            rows.setdefault(l[:34].strip(), []).append(l[34:].split()[-1])
print("%-34s " % "layer (ms / launch)" + " ".join("%9s" % f for f in forms))
for k, v in rows.items():
    print("%-34s " % k + " ".join("%9s" % t for t in v))
PY
for f in $forms; do
  echo -n "step MI355_WINO_FORM=$f: "
  MI355_WINO_FORM=$f python bench.py --no-cpu-baseline --no-precision-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], 'ms/step', d['value'], 'vol/s | ', {k:(round(v['s']*1e3/3,2), v['launches']//3) for k,v in list(r['all_kernels'].items())[:6]})"
done | tee $out/step.txt
