#!/bin/bash
# Round 5, last GPU call: what the driver runs at round end that tools/r5_host.sh did not -- smoke(), the two sliding-window cases, the
# DEFAULT bench line.  Usage: tools/r5_last.sh <tag>
tag=${1:-r5z}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 200 python -m pytest tests -m gpu -q -x -k "c5_sliding_window" > $out/pytest_c5.log 2>&1; tail -2 $out/pytest_c5.log
timeout 330 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json; tail -3 $out/bench.err
