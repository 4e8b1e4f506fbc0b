#!/bin/bash
# Runs on the GPU box: serialized kernel trace + HBM-traffic PMC passes of a bench.py configuration other than the default line.
#   tools/trace_bf16.sh <tag> [bench.py arguments]      default arguments: --precision bf16   (C3: --config c3)
# -> gpurun_out/<tag>/trace_bf16/bench_kernel_stats.csv, gpurun_out/<tag>/hbm_traffic_pmc_bf16.csv
export TMPDIR=/tmp
root=$(pwd)
out=gpurun_out/${1:-bf16}; mkdir -p $out
shift
args=${*:---precision bf16}
(cd /tmp && MI355_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $root/$out/trace_bf16 -o bench -- python $root/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/log.txt 2>&1)
rm -f $out/trace_bf16/bench_kernel_trace.csv
(cd /tmp && MI355_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $root/$out/pmc_fetch -o bench -- python $root/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/pmc_fetch.log 2>&1)
(cd /tmp && MI355_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $root/$out/pmc_write -o bench -- python $root/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/pmc_write.log 2>&1)
python tools/pmc_summary.py $out/pmc_fetch/bench_counter_collection.csv $out/pmc_write/bench_counter_collection.csv > $out/hbm_traffic_pmc_bf16.csv 2> $out/pmc_summary.err
rm -rf $out/pmc_fetch $out/pmc_write
head -30 $out/trace_bf16/bench_kernel_stats.csv | cut -c1-150
