#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py -m gpu -x -q > gpurun_out/r03v_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03v_tests.log | tail -2
(EXP_SHAPE=config2 timeout 600 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=0 debug_flags=0; timeout 600 python tools/exp_policy.py 2e7 debug_flags=0 debug_flags=0) > gpurun_out/r03v_time.txt 2>&1
cat gpurun_out/r03v_time.txt | cut -c1-150
