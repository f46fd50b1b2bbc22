#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python tools/exp_policy.py 1e8 debug_flags=0 log_capacity=3100000000 log_capacity=2000000000) > gpurun_out/r03ab_logcap.txt 2>&1
cat gpurun_out/r03ab_logcap.txt | cut -c1-170
