#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python tools/exp_policy.py 1e7 drain_split=1 drain_split=0 drain_split=1 drain_split=0; timeout 900 python tools/exp_policy.py 1e8 drain_split=1 drain_split=0; EXP_SHAPE=config2 timeout 900 python tools/exp_policy.py 1e7 drain_split=1 drain_split=0 drain_split=1 drain_split=0) > gpurun_out/r03l_split.txt 2>&1
cat gpurun_out/r03l_split.txt
