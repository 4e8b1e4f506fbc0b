#!/bin/bash
out=gpurun_out/r1c; mkdir -p $out
timeout 600 python tools/audit_ops.py --gpu --tc > $out/audit_unet_tc.log 2>&1; tail -14 $out/audit_unet_tc.log
timeout 600 python tools/audit_ops.py --gpu --model dynunet --filters 32,64,96 --dhw 16,24,32 > $out/audit_dyn.log 2>&1; tail -14 $out/audit_dyn.log
timeout 900 python -m pytest tests -m gpu -q -k "dynunet or wgrad_bf16 or transposed" > $out/pytest_sel.log 2>&1; tail -5 $out/pytest_sel.log
timeout 600 python tools/bench_ops.py bf16x3 bf16 > $out/bench_ops.log 2>&1; tail -30 $out/bench_ops.log
