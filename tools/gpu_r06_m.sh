#!/bin/bash
# Round 6, call 15: the walk's first hot sector requested in front of the refill's round trip (-DTMC_WALK_PREFETCH build) against the default
OUT=gpurun_out/r06_m; mkdir -p $OUT; export TMPDIR=/tmp
TARDIS_MC_LIB=$PWD/scratch/lib_pf.so timeout 900 python -m pytest tests/test_heavy_blocks.py tests/test_config3_shape.py tests/test_walk_hot_sectors.py -x -q > $OUT/pytest_pf.log 2>&1; echo "rc=$?" >> $OUT/pytest_pf.log
for rep in 1 2; do for v in base pf; do
  LIB=""; [ $v != base ] && LIB=$PWD/scratch/lib_$v.so
  echo "== $v rep $rep" >> $OUT/pf.log
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/pf.log 2>&1
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 ls_waves_per_simd=4 >> $OUT/pf.log 2>&1
done; done
tail -n 4 $OUT/pytest_pf.log; cat $OUT/pf.log
