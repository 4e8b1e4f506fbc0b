#!/bin/bash
out=gpurun_out/r6i
mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "wgrad" 2>&1 | tail -3
python tools/bench_stride2.py 2>&1 | grep -v amdgpu
root=$(pwd)
(cd /tmp && TMPDIR=/tmp MI355_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -f csv -d $root/$out/trace -o b -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-modes --no-c3 > $root/$out/trace.log 2>&1)
grep -E "wgrad_reduce|wgrad_wino_ring|wgrad_mfma" $out/trace/b_kernel_stats.csv | cut -c1-200
rm -rf $out/trace
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-precision-modes --no-c3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
