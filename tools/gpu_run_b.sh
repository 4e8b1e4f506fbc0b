#!/bin/bash
out=gpurun_out/r1d; mkdir -p $out
timeout 600 python tools/audit_ops.py --gpu --tc --fwd > $out/audit_unet_tc.log 2>&1; grep -E "<<<<|dice|proj|kernels" $out/audit_unet_tc.log | head -40
timeout 600 python tools/audit_ops.py --gpu --fwd --model dynunet --filters 32,64,96 --dhw 16,24,32 > $out/audit_dyn.log 2>&1; grep -E "<<<<|dice|proj|kernels" $out/audit_dyn.log | head -40
