#!/bin/bash
mkdir -p gpurun_out
TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_timers.so timeout 900 python tools/sections_cfg3.py 5e5 debug_flags=0 > gpurun_out/r03q_sections_drain.txt 2>&1
cat gpurun_out/r03q_sections_drain.txt
