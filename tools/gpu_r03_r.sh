#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "config5" > gpurun_out/r03r_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03r_tests.log | tail -2; grep -n "Error\|assert" gpurun_out/r03r_tests.log | head
