#!/bin/bash
# First GPU call of round 4 (≈13 min of box time): what round 3 could not collect after its last kernel landed.
#   gpurun --timeout 1500 -- 'tools/r4_first_call.sh'
# 1. the FULL GPU suite with the plane-ring 16-bit kernel as the default route (round 3 ran only its op twins and the bf16 / fp16 network tests)
# 2. fp32 bench line + serialized kernel trace + PMC traffic (tools/gpu_profile.sh)
# 3. the bf16 step: kernel trace + PMC traffic (tools/trace_bf16.sh), tile vs plane-ring whole-step A/B, C3 (batch 4) A/B
# 4. SQ counters of the plane-ring kernel against the tile kernel on the 32 -> 32 @128^3 layer
tools/gpu_profile.sh r4 ${R4_WHAT:-bench trace pmc}   # the full suite already ran green on this default in GPUTEST_r03 (driver, head 5eee73d)
tools/trace_bf16.sh r4bf16
out=gpurun_out/r4; mkdir -p $out
for f in tile auto; do
  echo -n "bf16 step MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --precision bf16 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  echo -n "c3 (bf16, batch 4) MI355_BF16_FORM=$f: "; MI355_BF16_FORM=$f python bench.py --config c3 --no-cpu-baseline --no-precision-modes --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $out/bf16_form_ab.txt
MI355_BF16_FORM=tile tools/sq_counters.sh r4sq_tile "bf16 32 32 128 fwdplain" "bf16 32 32 128 fwd"
MI355_BF16_FORM=auto tools/sq_counters.sh r4sq_zring "bf16 32 32 128 fwdplain" "bf16 32 32 128 fwd"
