#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_volley_queue.py tests/test_vpacket_screening.py -m gpu -x -q > gpurun_out/r03t_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03t_tests.log | tail -2
(timeout 1200 python tools/exp_vpk.py config5 1e7 variant=-1 variant=4 variant=4,vq_min_items=0) > gpurun_out/r03t_vq.txt 2>&1
cat gpurun_out/r03t_vq.txt | cut -c1-200
