#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03y_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03y_tests.log | tail -2; grep -n "^FAILED\|Error" gpurun_out/r03y_tests.log | head
