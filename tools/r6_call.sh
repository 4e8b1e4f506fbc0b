#!/bin/bash
# scratch per-call script (GPU box)
mkdir -p gpurun_out/gn
for args in "fp32 2" "bf16 4"; do for v in 1 0; do
(cd /tmp && export TMPDIR=/tmp && MI355_GN_APPLY_ROWS=$v timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/gn/t$v -o b -- python $GRAFT_REPO_ROOT/tools/bench_gn_bwd.py $args 2>&1 | grep "ch @")
f=$(find gpurun_out/gn/t$v -name "*kernel_stats.csv" | head -1); echo "== $args rows=$v"; grep "gn_bwd" $f | awk -F'",' '{print substr($1,1,50), $2}'
rm -rf gpurun_out/gn/t$v
done; done
