#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python tools/exp_vpk.py config5 2e6 debug_flags=0 blocks_per_cu=3 blocks_per_cu=5 blocks_per_cu=6 group_size=8 variant=2) > gpurun_out/r03h_vpk.txt 2>&1
cat gpurun_out/r03h_vpk.txt
