#!/bin/bash
# Round 6: counters behind "the accumulate kernel is bound by instruction issue": both forms in one process, SQ counters per kernel
OUT=gpurun_out/r06_ac; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
cd /tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  EXP_LEVELS=heavy timeout -k 5 600 rocprofv3 --pmc $c -d $ROOT/$OUT/pmc_$i -o pmc -- python $ROOT/tools/exp_cfg3.py 4e7 est_accumulate=2,log_sets=1 est_accumulate=3,log_sets=1 > $ROOT/$OUT/pmc_$i.log 2>&1
  tail -n 2 $ROOT/$OUT/pmc_$i.log | cut -c1-200
done
cd $ROOT
python tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -E "accumulate_dyadic" $OUT/summary.txt | cut -c1-200
find $OUT -name "*.db" -delete
