#!/bin/bash
# Re-collection after the last source edits (comments + mi355_conv3d_wino_supported): sanity tests of the rebuilt kernels, then tools/r4_final.sh
out=gpurun_out/r4f; mkdir -p $out
timeout 900 python -m pytest tests/test_wino_gpu.py tests/test_ops_gpu.py tests/test_oracle_pinned.py tests/test_graph.py -m gpu -q -x > $out/pytest_sanity.log 2>&1; echo "pytest rc=$?" >> $out/pytest_sanity.log; tail -3 $out/pytest_sanity.log
tools/r4_final.sh
