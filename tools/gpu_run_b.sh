#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_input_channels.py -m gpu -q 2>&1 | grep -E "^E|passed|failed" | head
