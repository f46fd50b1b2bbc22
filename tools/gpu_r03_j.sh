#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py -m gpu -x -q > gpurun_out/r03j_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03j_tests.log | tail -2
(timeout 900 python tools/exp_policy.py 4e7 debug_flags=0 debug_flags=0; timeout 900 python tools/exp_policy.py 4e7 heavy debug_flags=0) > gpurun_out/r03j_pack.txt 2>&1
cat gpurun_out/r03j_pack.txt
