#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_volley_queue.py tests/test_config3_shape.py -m gpu -x -q > gpurun_out/r03e_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03e_tests.log | tail -2
(timeout 900 python tools/exp_vpk.py config5 1e6 debug_flags=0 debug_flags=33554432; timeout 600 python tools/exp_vpk.py config2v 2e6 debug_flags=0 debug_flags=33554432 variant=1 variant=1,debug_flags=33554432) > gpurun_out/r03e_vpk.txt 2>&1
cat gpurun_out/r03e_vpk.txt
