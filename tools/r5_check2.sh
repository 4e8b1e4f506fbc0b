#!/bin/bash
tag=${1:-r5o}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_headline_parity_gpu.py tests/test_dynunet.py tests/test_input_channels.py -m gpu -q --durations=5 --deselect tests/test_model_gpu.py::test_train_mode_loss_trajectory_matches_oracle > $out/pytest_rest.log 2>&1; tail -12 $out/pytest_rest.log
