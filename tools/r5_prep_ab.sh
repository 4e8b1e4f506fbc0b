#!/bin/bash
# ON BRANCH r5-prep. Before the call (here, CPU): python 3dunetcnn_amd/build.py
#   tools/build_variant.sh occ1 conv3d_wgrad_bf16.hip -DWGRAD_LP_WAVES=1 ; tools/build_variant.sh nolean conv3d_wgrad_bf16.hip -DWGRAD_LP_NO_LEAN
#   git stash / worktree of main -> tools/build_variant.sh head   (tools/libvar_head.so = main's kernels)
# On the GPU box (≈5 min): storage tests on the branch kernels, the weight-gradient variants layer by layer, the bf16 / C3 step of the branch
# build against main's library (tools/ab_lib.py), i.e. the two unmeasured changes of the branch: 16-byte streaming path, 72-register wgrad.
out=gpurun_out/r5p; mkdir -p $out
timeout 600 python -m pytest tests/test_act_storage_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "act_storage or wgrad_bf16" > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt; tail -2 $out/tests.txt
export MI355_PRECISION=bf16
for c in "32 32 128" "64 32 128" "64 64 64" "128 128 32"; do
  echo -n "wgrad bf16 $c |"
  for lib in tools/libvar_head.so tools/libvar_nolean.so tools/libvar_occ1.so ""; do
    echo -n " ${lib:-branch}: "; ONE_CONV_LIB=$lib python tools/bench_wgrad.py $c 2>/dev/null | tail -1 | tr '\n' ' '
  done; echo
done | tee $out/wgrad_layers.txt
for c in "64 64 64" "128 128 32"; do echo -n "wgrad bf16 $c, 32-channel workgroups everywhere (branch): "; MI355_WGRAD_LP_MT=1 python tools/bench_wgrad.py $c 2>/dev/null | tail -1; done | tee -a $out/wgrad_layers.txt
MI355_STORAGE=bf16 python tools/ab_lib.py tools/libvar_head.so 2 2>/dev/null | tee $out/step_ab.txt
