#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vpacket_screening.py tests/test_hip_parity.py tests/test_volley_queue.py tests/test_config3_shape.py -m gpu -x -q > gpurun_out/r03p_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03p_tests.log | tail -2
(timeout 900 python tools/exp_vpk.py config5 3e6 variant=-1 variant=1; timeout 900 python tools/exp_vpk.py config5 1e7 variant=-1 variant=1) > gpurun_out/r03p_vpk.txt 2>&1
cat gpurun_out/r03p_vpk.txt | cut -c1-200
