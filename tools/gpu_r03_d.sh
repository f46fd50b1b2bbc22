#!/bin/bash
mkdir -p gpurun_out
(echo "== live-lane-scaled cut-offs"; timeout 900 python tools/exp_drain.py 1e6 1e7 2e7 4e7; echo "== fixed cut-offs (debug flag 16777216)"; EXP_BASE_FLAGS=16777216 timeout 900 python tools/exp_drain.py 1e6 1e7 2e7) > gpurun_out/r03d_drain.txt 2>&1
cat gpurun_out/r03d_drain.txt
EXP_SHAPE=config2 timeout 600 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=16777216 debug_flags=0 > gpurun_out/r03d_cfg2.txt 2>&1
cat gpurun_out/r03d_cfg2.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py -m gpu -x -q 2>&1 | tail -3
