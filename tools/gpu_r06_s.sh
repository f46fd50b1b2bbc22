#!/bin/bash
# Round 6: accumulate_dyadic_kernel<.., LOOP> (est_accumulate 3, a lane per record) -- parity, then kernel times against est_accumulate 2
OUT=gpurun_out/r06_s; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_estimator_pipelines.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
cd /tmp
EXP_LEVELS=heavy timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o t -- python $ROOT/tools/exp_cfg3.py 4e7 est_accumulate=2 est_accumulate=3 > $ROOT/$OUT/trace_run.txt 2>&1
cd $ROOT
cat $OUT/trace_run.txt | tail -n 3
python tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1; grep -E "accumulate|partition|propagate_wave|bin_" $OUT/summary.txt | head -12
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 est_accumulate=2 est_accumulate=3 est_accumulate=2 est_accumulate=3 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt
