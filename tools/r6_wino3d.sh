#!/bin/bash
# Round 6, first call for conv3d_wino3d: parity of both Winograd forward kernels on the GPU, then the per-layer A/B on one box.
out=gpurun_out/r6a
mkdir -p $out
python -m pytest tests/test_wino_gpu.py -q -x 2>&1 | tail -5 > $out/pytest_wino.txt
MODES=plain,norm+moments,plain+gnb python tools/bench_conv_layers.py tree@2d tree@3d > $out/layers.txt 2>&1
tail -40 $out/layers.txt
cat $out/pytest_wino.txt
