#!/bin/bash
mkdir -p gpurun_out
EXP_SHAPE=config5 TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_timers.so timeout 900 python tools/sections_cfg3.py 2e6 variant=2 > gpurun_out/r03n_sections_cfg5.txt 2>&1
cat gpurun_out/r03n_sections_cfg5.txt
