#!/bin/bash
# Runs on the GPU box (via gpurun): everything needed to decide on landing the Winograd kernels, in one call (~4 min of box time).
#   tools/wino_landing.sh [tag]    -> gpurun_out/<tag>/{tests.txt, layers_*.txt, wgrad.txt, step.txt}
tag=${1:-wino}; out=gpurun_out/$tag; mkdir -p $out
# 1. parity on hardware: the GPU twins of the emulator tests (all three loop forms, whole-network step incl. wgrad)
timeout 900 python -m pytest tests/test_wino_gpu.py -q -m gpu > $out/tests.txt 2>&1; tail -3 $out/tests.txt
# 2. forward / dgrad per layer: direct kernel vs the three Winograd loop forms
MI355_WINOGRAD=0 python tools/bench_conv_layers.py > $out/layers_direct.txt 2>&1
MI355_WINOGRAD=1 MI355_WINO_PIPE=1 MI355_WINO_BMODE=2 python tools/bench_conv_layers.py > $out/layers_wino_b2.txt 2>&1      # weights one use ahead (default)
MI355_WINOGRAD=1 MI355_WINO_PIPE=1 MI355_WINO_BMODE=0 python tools/bench_conv_layers.py > $out/layers_wino_b0.txt 2>&1      # the form measured in round 2
MI355_WINOGRAD=1 MI355_WINO_PIPE=1 MI355_WINO_BMODE=1 python tools/bench_conv_layers.py > $out/layers_wino_b1.txt 2>&1      # weights first, transform under their latency
MI355_WINOGRAD=1 MI355_WINO_PIPE=0 python tools/bench_conv_layers.py > $out/layers_wino_3barrier.txt 2>&1
echo "layer | direct | wino BMODE 2 | BMODE 0 | BMODE 1 | three-barrier"
paste $out/layers_direct.txt $out/layers_wino_b2.txt $out/layers_wino_b0.txt $out/layers_wino_b1.txt $out/layers_wino_3barrier.txt | grep -E "k3|sum" | cut -c1-50,85-100,135-150,185-200,235-250
# 3. weight gradient per layer: ring kernel vs Winograd (pipelined / three-barrier)
for c in "32 32 128" "64 64 64" "128 128 32" "256 256 16" "64 32 128"; do
  for v in "0 1" "1 1" "1 0"; do set -- $v
    echo -n "wgrad $c  MI355_WINOGRAD_WGRAD=$1 MI355_WINO_PIPE=$2: "
    MI355_WINOGRAD_WGRAD=$1 MI355_WINO_PIPE=$2 python tools/one_conv.py fp32 $c wgrad 12 2>/dev/null | tail -1
  done
done | tee $out/wgrad.txt
# 4. whole step: product path, forward/dgrad on Winograd, forward/dgrad/wgrad on Winograd
for v in "0 0" "1 0" "1 1"; do set -- $v
  echo -n "step MI355_WINOGRAD=$1 MI355_WINOGRAD_WGRAD=$2: "
  MI355_WINOGRAD=$1 MI355_WINOGRAD_WGRAD=$2 python bench.py --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | cut -c60-175
done | tee $out/step.txt
