#!/bin/bash
# scratch per-call script (GPU box)
echo "== colsPer 1 (tree)"; python tools/bench_wgrad_lp.py 2>&1 | grep -v amdgpu | head -10
echo "== MI355_WGRAD_LP_COLS=2"; MI355_WGRAD_LP_COLS=2 python tools/bench_wgrad_lp.py 2>&1 | grep -v amdgpu | head -10
echo "== MI355_WGRAD_LP_COLS=4"; MI355_WGRAD_LP_COLS=4 python tools/bench_wgrad_lp.py 2>&1 | grep -v amdgpu | head -10
