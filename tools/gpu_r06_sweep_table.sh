#!/bin/bash
# Round 6, item 1: the interleaved {nu, tau} sweep table (option sweep_table 0 / 1 / 2) -- parity first, then A/B in one process per workload.
OUT=gpurun_out/r06_sweep_table; mkdir -p $OUT
timeout 900 python -m pytest tests/test_round6_options.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="sweep_table=0 sweep_table=1 sweep_table=2 sweep_table=0 sweep_table=1 sweep_table=2"
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 $E > $OUT/heavy_2e7.log 2>&1
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/heavy_1e8.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 sweep_table=0 sweep_table=1 sweep_table=2 > $OUT/uniform_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 sweep_table=0 sweep_table=1 sweep_table=2 ls_waves_per_simd=3,sweep_table=0 ls_waves_per_simd=3,sweep_table=1 > $OUT/heavy_1.25e7.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/config2_1e7.log 2>&1
tail -n 30 $OUT/*.log
