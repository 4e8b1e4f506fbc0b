#!/bin/bash
# Round 5: the GPU tests that exercise the Winograd forward / dgrad kernel beyond its op tests (per-launch fp64 audit of the headline step,
# whole-network tests, full-size adjoint / agreement tests), then the default bench line.  Usage: tools/r5_check.sh <tag>
tag=${1:-r5n}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_wino_gpu.py tests/test_launch_audit.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -x --durations=8 > $out/pytest_wino_paths.log 2>&1; tail -15 $out/pytest_wino_paths.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
