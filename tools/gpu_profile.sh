#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test suite, bench line, rocprofv3 kernel trace + HBM-traffic PMC passes.
# Usage: tools/gpu_profile.sh <tag> [what...]   what in: tests bench trace pmc   (default: all)
# trace / pmc run bench.py with MI355_SIDE_STREAM=0: every kernel alone on one stream, so that a launch's duration (and its counters)
# are its own -- the same serialized form bench.py's `roofline` block is measured in. `bench` is the default command (overlap on).
tag=${1:-r2}; shift
what=${*:-tests bench trace pmc}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
for w in $what; do
case $w in
tests) timeout 1800 python -m pytest tests -m gpu -q -s --durations=30 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log;;
bench) timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json;;
trace) (cd /tmp && MI355_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $root/$out/trace -o bench -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-modes > $root/$out/trace.log 2>&1)
  rm -f $out/trace/bench_kernel_trace.csv; ls $out/trace | head;;
pmc)
  (cd /tmp && MI355_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $root/$out/pmc_fetch -o bench -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/pmc_fetch.log 2>&1)
  (cd /tmp && MI355_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $root/$out/pmc_write -o bench -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/pmc_write.log 2>&1)
  (cd /tmp && MI355_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -f csv -d $root/$out/pmc_clk -o bench -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/pmc_clk.log 2>&1)
  python tools/pmc_summary.py $out/pmc_fetch/bench_counter_collection.csv $out/pmc_write/bench_counter_collection.csv $out/pmc_clk/bench_counter_collection.csv > $out/hbm_traffic_pmc.csv 2> $out/pmc_summary.err
  head -4 $out/hbm_traffic_pmc.csv
  rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_clk;;
esac
done
# keep the merge-back small: drop anything above 20 MB
find $out -size +20M -delete
du -sh $out
