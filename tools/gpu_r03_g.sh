#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vpacket_screening.py tests/test_hip_parity.py tests/test_volley_queue.py tests/test_boundary_gpu.py -m gpu -x -q > gpurun_out/r03g_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03g_tests.log | tail -2
grep -n "Error\|assert" gpurun_out/r03g_tests.log | head -20
(timeout 900 python tools/exp_vpk.py config5 3e6 debug_flags=0 vpacket_screening=0) > gpurun_out/r03g_vpk.txt 2>&1
cat gpurun_out/r03g_vpk.txt
