#!/bin/bash
# GPU call 2 of round 4: the new GPU tests (16-bit launch audit incl. C3 at 128^3 x 4, graph + second model), the C3 bench line with its
# roofline block, the serialized C3 trace + PMC traffic, per-layer bf16 launch times (baseline of the kernel work).
out=gpurun_out/r4c3; mkdir -p $out
timeout 900 python -m pytest tests/test_launch_audit.py tests/test_graph.py -m gpu -q -s --durations=10 > $out/pytest_new.log 2>&1; echo "pytest rc=$?" >> $out/pytest_new.log; tail -12 $out/pytest_new.log
timeout 600 python bench.py --config c3 > $out/bench_c3.json 2> $out/bench_c3.err; tail -c 1500 $out/bench_c3.json
tools/trace_bf16.sh r4c3 --config c3 > $out/trace.log 2>&1
PREC=bf16 timeout 300 python tools/bench_conv_layers.py > $out/bf16_layers.txt 2>&1; cat $out/bf16_layers.txt
