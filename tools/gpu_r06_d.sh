#!/bin/bash
# Round 6, call 4: the shell-sorted log with LDS-atomic slots -- parity, then A/B
OUT=gpurun_out/r06_d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_estimator_pipelines.py tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py tests/test_round6_options.py tests/test_round5_options.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="log_sets=1,log_by_shell=0 log_sets=1,log_by_shell=1 log_sets=1,log_by_shell=0 log_sets=1,log_by_shell=1 log_by_shell=0 log_by_shell=1"
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 $E > $OUT/sl_heavy_2e7.log 2>&1
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 log_by_shell=0 log_by_shell=1 log_by_shell=0 log_by_shell=1 > $OUT/sl_heavy_1e8.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 log_by_shell=0 log_by_shell=1 > $OUT/sl_uniform_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 log_by_shell=0,ls_waves_per_simd=3 log_by_shell=1,ls_waves_per_simd=3 log_by_shell=0 log_by_shell=1 > $OUT/sl_heavy_1.25e7.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/sl_config2_1e7.log 2>&1
HOST_PIPE=1 timeout 600 python tools/time_boundary.py 1e7 3 > $OUT/boundary_config3_pipe1.log 2>&1
HOST_PIPE=0 timeout 600 python tools/time_boundary.py 1e7 3 > $OUT/boundary_config3_pipe0.log 2>&1
HOST_PIPE=1 timeout 600 python tools/time_boundary.py 1e7 2 > $OUT/boundary_config2_pipe1.log 2>&1
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_one_shot_run.py tests/test_full_size_configs.py -x -q > $OUT/pytest_boundary.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_boundary.log
tail -n 40 $OUT/*.log
