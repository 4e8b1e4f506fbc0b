export TMPDIR=/tmp
root=$(pwd)
for cfg in "0 0" "0 1"; do
set -- $cfg
out=gpurun_out/r2g/side$1_fused$2; mkdir -p $out
(cd /tmp && MI355_SIDE_STREAM=$1 MI355_FUSED_STATS=$2 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $root/$out -o bench -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-modes --no-kernel-events > $root/$out/log.txt 2>&1)
rm -f $out/bench_kernel_trace.csv
done
ls gpurun_out/r2g/*
