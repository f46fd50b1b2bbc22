#!/bin/bash
# Round 6, call 10: ten / twelve lines per step in the sixteen-wave instantiation on the interleaved table (builds with -DTMC_LS_CHUNK_NT=10 / 12 under scratch/)
OUT=gpurun_out/r06_j; mkdir -p $OUT; export TMPDIR=/tmp
for ch in 10 12; do
  TARDIS_MC_LIB=$PWD/scratch/lib_ch$ch.so timeout 900 python -m pytest tests/test_round6_options.py tests/test_config3_shape.py tests/test_heavy_blocks.py -x -q > $OUT/pytest_ch$ch.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ch$ch.log
done
for rep in 1 2; do
for ch in 8 10 12; do
  LIB=""; [ $ch != 8 ] && LIB=$PWD/scratch/lib_ch$ch.so
  echo "== CH=$ch rep $rep" >> $OUT/heavy_1e8.log
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/heavy_1e8.log 2>&1
done; done
for ch in 8 10 12; do
  LIB=""; [ $ch != 8 ] && LIB=$PWD/scratch/lib_ch$ch.so
  echo "== CH=$ch" >> $OUT/others.log
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 ls_waves_per_simd=4 ls_waves_per_simd=4 >> $OUT/others.log 2>&1
  TARDIS_MC_LIB=$LIB timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/others.log 2>&1
  TARDIS_MC_LIB=$LIB EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 ls_waves_per_simd=4 ls_waves_per_simd=4 >> $OUT/others.log 2>&1
done
tail -n 5 $OUT/pytest_ch*.log; cat $OUT/heavy_1e8.log $OUT/others.log
