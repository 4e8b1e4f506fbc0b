#!/bin/bash
# Runs on the GPU box: the z-marching Winograd forward kernel (MI355_WINO_FORM=zring / auto) against the tile form, per layer and whole step.
tag=${1:-zring_ab}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_wino_gpu.py -q -m gpu -k "z-marching" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
for f in tile zring auto; do MI355_WINO_FORM=$f python tools/bench_conv_layers.py > $out/layers_$f.txt 2>&1; done
python - <<PY | tee $out/layers.txt
rows = {}
for f in ("tile", "zring", "auto"):
    for l in open("$out/layers_%s.txt" % f):
        if " k3 " in l or l.startswith("sum"):
            rows.setdefault(l[:34].strip(), []).append(l[34:].split()[-1])
print("%-34s %9s %9s %9s" % ("layer (ms / launch)", "tile", "zring", "auto"))
for k, v in rows.items():
    print("%-34s " % k + " ".join("%9s" % t for t in v))
PY
for f in tile auto; do
  echo -n "step MI355_WINO_FORM=$f: "
  MI355_WINO_FORM=$f python bench.py --no-cpu-baseline --no-precision-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], 'ms/step', d['value'], 'vol/s | ', {k:(round(v['s']*1e3/3,2), v['launches']//3) for k,v in list(r['all_kernels'].items())[:6]})"
done | tee $out/step.txt
