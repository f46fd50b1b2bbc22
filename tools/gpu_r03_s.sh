#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=134217728 debug_flags=0 debug_flags=134217728; timeout 900 python tools/exp_policy.py 1e6 debug_flags=0 debug_flags=134217728; EXP_SHAPE=config2 timeout 900 python tools/exp_policy.py 1e7 debug_flags=0 debug_flags=134217728) > gpurun_out/r03s_ahead.txt 2>&1
cat gpurun_out/r03s_ahead.txt | cut -c1-170
