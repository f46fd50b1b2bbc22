#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python tools/exp_policy.py 1e8 debug_flags=0 debug_flags=1 log_sets=1) > gpurun_out/r03x_noest.txt 2>&1
cat gpurun_out/r03x_noest.txt | cut -c1-170
