#!/bin/bash
# Round 6, call 18: the straight-line (branch-free) form of the lean sweep in the sixteen-wave instantiation (-DTMC_STRAIGHT_A=1 build) against the per-line loop
OUT=gpurun_out/r06_n; mkdir -p $OUT; export TMPDIR=/tmp
TARDIS_MC_LIB=$PWD/scratch/lib_straight.so timeout 900 python -m pytest tests/test_round6_options.py tests/test_config3_shape.py tests/test_heavy_blocks.py -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
for rep in 1 2; do for v in base straight; do
  LIB=""; [ $v != base ] && LIB=$PWD/scratch/lib_$v.so
  echo "== $v rep $rep" >> $OUT/ab.log
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/ab.log 2>&1
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 ls_waves_per_simd=4 >> $OUT/ab.log 2>&1
  TARDIS_MC_LIB=$LIB timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/ab.log 2>&1
  TARDIS_MC_LIB=$LIB EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 ls_waves_per_simd=4 >> $OUT/ab.log 2>&1
done; done
tail -n 4 $OUT/pytest.log; cat $OUT/ab.log
