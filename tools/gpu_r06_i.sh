#!/bin/bash
# Round 6, call 9: the lean proof in the twelve-line (straight-line) instantiation: parity, then A/B against the sixteen-wave one
OUT=gpurun_out/r06_i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_round6_options.py tests/test_round5_options.py tests/test_config3_shape.py tests/test_heavy_blocks.py tests/test_boundary_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="ls_waves_per_simd=4 ls_waves_per_simd=3,sweep_table=0 ls_waves_per_simd=3,sweep_table=1 ls_waves_per_simd=4 ls_waves_per_simd=3,sweep_table=0 ls_waves_per_simd=3,sweep_table=1"
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/b_lean_heavy_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 $E > $OUT/b_lean_heavy_1.25e7.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/b_lean_heavy_1e7.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 ls_waves_per_simd=3,sweep_table=0 ls_waves_per_simd=3,sweep_table=1 > $OUT/b_lean_uniform_1e8.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/b_lean_config2_1e7.log 2>&1
BOUNDARY_MODE=macroatom timeout 600 python tools/time_boundary.py 4e4 1 > $OUT/boundary_tardis_example.log 2>&1
tail -n 12 $OUT/pytest.log; tail -n 8 $OUT/b_lean*.log; tail -n 14 $OUT/boundary_tardis_example.log
