#!/bin/bash
# on the GPU box: FETCH_SIZE / WRITE_SIZE of the four 1 GiB access patterns of tools/micro/pmc_calibrate.hip (separate passes)
export TMPDIR=/tmp; root=$(pwd); out=gpurun_out/micro; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  d=$root/$out/cal_$c; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $d -o cal -- $root/tools/micro/pmc_calibrate > /dev/null 2>&1)
  python tools/pmc_kernel.py $d/cal_counter_collection.csv "_f" ; rm -rf $d
done | tee $out/pmc_calibrate.txt
