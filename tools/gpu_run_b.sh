#!/bin/bash
# scratch: seeds sweep of the transposed-conv variant gradient parity
cd /root/repo
cat > /tmp/sweep.py <<'PY'
import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_model_gpu as T
for seed in (1234, 1, 2, 3, 4, 5):
    e = T._run_pair(dict(n_features=4, n_outputs=3, use_transposed_convolutions=True), (1, 2, 2, 4), (32, 32, 32), 1, tc=True, seed=seed)
    print(seed, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in e.items()}, flush=True)
PY
for env in "X=1" "MI355_GN_VPB=2048"; do
  echo "=== $env"
  env $env timeout 900 python /tmp/sweep.py 2>&1 | tail -8 | cut -c1-330
done
