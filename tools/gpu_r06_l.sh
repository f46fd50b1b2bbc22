#!/bin/bash
# Round 6, call 14: partition staging of 4096 records (one workgroup per CU) against 2048 (two); configs[4]'s own 6.25e7 packets for the record
OUT=gpurun_out/r06_l; mkdir -p $OUT; export TMPDIR=/tmp
TARDIS_MC_LIB=$PWD/scratch/lib_part4096.so timeout 900 python -m pytest tests/test_estimator_pipelines.py -x -q > $OUT/pytest_p4096.log 2>&1; echo "rc=$?" >> $OUT/pytest_p4096.log
for rep in 1 2; do for v in 2048 4096; do
  LIB=""; [ $v != 2048 ] && LIB=$PWD/scratch/lib_part$v.so
  echo "== PART_RECORDS=$v rep $rep" >> $OUT/part.log
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 log_sets=1,ls_waves_per_simd=4 >> $OUT/part.log 2>&1
  TARDIS_MC_LIB=$LIB EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1e8 ls_waves_per_simd=4 >> $OUT/part.log 2>&1
done; done
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > $OUT/bench_config5_full.json 2> $OUT/bench_config5_full.err
tail -n 4 $OUT/pytest_p4096.log; cat $OUT/part.log; python -c "
import json; d=json.loads(open('$OUT/bench_config5_full.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['workload'][:80])"
