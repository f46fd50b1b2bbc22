#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config3_shape.py -m gpu -x -q > gpurun_out/r03aa_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03aa_tests.log | tail -2
(echo "== fp32 rounded-down frequencies in the lane sweeps"; timeout 600 python tools/exp_policy.py 4e7 debug_flags=0; EXP_SHAPE=config2 timeout 600 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=0
 echo "== fp64 frequencies (previous build)"; TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_fp64nu.so timeout 600 python tools/exp_policy.py 4e7 debug_flags=0; EXP_SHAPE=config2 TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_fp64nu.so timeout 600 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=0
 echo "== fp32 again"; timeout 600 python tools/exp_policy.py 4e7 debug_flags=0) > gpurun_out/r03aa_nu32.txt 2>&1
cat gpurun_out/r03aa_nu32.txt | cut -c1-150
