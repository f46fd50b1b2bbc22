#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python tools/exp_vpk.py config5 3e6 variant=1 variant=3 variant=2) > gpurun_out/r03m_cfg5b.txt 2>&1
cat gpurun_out/r03m_cfg5b.txt | cut -c1-220
