#!/bin/bash
# scratch per-call script (GPU box)
export ONE_CONV_LIB=tools/libvar_j4.so LAYERS="256,256,16;512,256,16;128,256,16"
echo "== J = 4 (64-channel chunks) on the small wide form"; python tools/bench_lp_tile.py 2>&1 | grep -v amdgpu
echo "== MI355_LP_J4=0"; MI355_LP_J4=0 python tools/bench_lp_tile.py 2>&1 | grep -v amdgpu
