#!/bin/bash
mkdir -p gpurun_out
(echo "== 3 waves per SIMD (168 VGPRs)"; timeout 900 python tools/exp_vpk.py config5 3e6 variant=2; timeout 600 python tools/exp_vpk.py config2v 2e6 variant=2
 echo "== 2 waves per SIMD (256 VGPRs)"; TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_w2.so timeout 900 python tools/exp_vpk.py config5 3e6 variant=2; TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_w2.so timeout 600 python tools/exp_vpk.py config2v 2e6 variant=2) > gpurun_out/r03o_w2.txt 2>&1
cat gpurun_out/r03o_w2.txt | cut -c1-200
