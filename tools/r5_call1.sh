#!/bin/bash
# Round 5, GPU call 1 (one box, ~10 min): (1) what ds_read_b64_tr_b16 hands each lane; (2) SQ counters of the 16-bit weight gradient and a
# FRESH set for the fp32 Winograd kernels (verdict item 1c); (3) the A/B of the two changes prepared on r5-prep (16-byte streaming path,
# 72-register 16-bit weight gradient) against main's round-4 library (tools/libvar_head.so).
out=gpurun_out/r5a; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe 2> $out/tr_read_build.log && tools/micro/tr_read_probe > $out/tr_read_probe.txt 2>&1
head -30 $out/tr_read_probe.txt
tools/sq_counters.sh r5a "bf16 32 32 128 wgrad" "bf16 64 64 64 wgrad" "fp32 32 32 128 fwdnormmom" "fp32 64 32 128 fwd" "fp32 256 256 32 fwdnormmom" "fp32 32 32 128 wgrad"
cp gpurun_out/r5a/sq_counters_conv_kernels.txt $out/sq_all.txt
bash tools/r5_prep_ab.sh
