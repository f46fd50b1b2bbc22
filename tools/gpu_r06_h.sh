#!/bin/bash
# Round 6, call 8: lane sweeps under v-packets (configs[4] shape), cut-offs, the stages of a tardis_example-sized drop-in call
OUT=gpurun_out/r06_h; mkdir -p $OUT; export TMPDIR=/tmp
EXP_SHAPE=config5 EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e7 variant=-1 variant=3,vpk_wide_registers=2 variant=3,vpk_wide_registers=0 variant=-1 variant=3,vpk_wide_registers=2 > $OUT/vpk_ls_config5.log 2>&1
EXP_SHAPE=config2v timeout 600 python tools/exp_cfg3.py 4e6 variant=-1 variant=3 variant=3,vpk_wide_registers=2 variant=-1 > $OUT/vpk_ls_config2v.log 2>&1
E="lane_sweep_min_active=12,walk_min_active=16 lane_sweep_min_active=12,walk_min_active=20 lane_sweep_min_active=12,walk_min_active=24 lane_sweep_min_active=16,walk_min_active=20 lane_sweep_min_active=12,walk_min_active=12 lane_sweep_min_active=12,walk_min_active=16"
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/cutoffs_heavy_1e8.log 2>&1
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1.25e7 lane_sweep_min_active=12,walk_min_active=12 lane_sweep_min_active=12,walk_min_active=16 lane_sweep_min_active=12,walk_min_active=20 > $OUT/cutoffs_heavy_1.25e7.log 2>&1
BOUNDARY_MODE=macroatom timeout 600 python tools/time_boundary.py 4e4 1 > $OUT/boundary_tardis_example.log 2>&1
tail -n 40 $OUT/*.log
