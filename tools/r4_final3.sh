#!/bin/bash
# Final collection of round 4 after the 16-bit activation storage landed: smoke, sanity tests of the retyped fp32 kernels (the storage tests
# and the audited 16-bit steps ran in tools/r4_storage_call.sh on the same sources), then tools/r4_final.sh.
out=gpurun_out/r4f; mkdir -p $out
python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 400 python -m pytest tests/test_wino_gpu.py tests/test_ops_gpu.py tests/test_oracle_pinned.py tests/test_graph.py -m gpu -q -x -k "not zring and not bf16_paths" > $out/pytest_sanity.log 2>&1; echo "pytest rc=$?" >> $out/pytest_sanity.log; tail -2 $out/pytest_sanity.log
timeout 200 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "autocast_unet_mixed_precision" >> $out/pytest_sanity.log 2>&1; echo "pytest rc=$?" >> $out/pytest_sanity.log; tail -2 $out/pytest_sanity.log
tools/r4_final.sh
