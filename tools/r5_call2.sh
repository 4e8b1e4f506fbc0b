#!/bin/bash
# Round 5, GPU call 2 (the first of the re-created container; ~7 min): state check of the tree (smoke), what ds_read_b64_tr_b16 hands each
# lane, a FRESH set of SQ counters for the fp32 Winograd kernels (verdict item 1c: the "before" of this round), the per-layer launch times
# of the step's conv layers and the default bench line.
out=gpurun_out/r5b; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe 2> $out/tr_read_build.log && timeout 60 tools/micro/tr_read_probe > $out/tr_read_probe.txt 2>&1
head -12 $out/tr_read_probe.txt
tools/sq_counters.sh r5b "fp32 32 32 128 fwdnormmom" "fp32 256 256 32 fwdnormmom" "fp32 32 32 128 wgrad"
timeout 300 python tools/bench_conv_layers.py > $out/conv_layers.txt 2>&1; tail -32 $out/conv_layers.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
