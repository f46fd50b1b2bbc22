#!/bin/bash
# Round 6: segment-tree cell layout of the dyadic accumulators -- parity of every accumulate mode, kernel times
OUT=gpurun_out/r06_y; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 1200 python -m pytest tests/test_estimator_pipelines.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
cd /tmp
EXP_LEVELS=heavy timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o t -- python $ROOT/tools/exp_cfg3.py 1e8 est_accumulate=3 > $ROOT/$OUT/trace_run.txt 2>&1
cd $ROOT
tail -n 1 $OUT/trace_run.txt
python tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1; grep -E "accumulate|partition|propagate_wave|bin_" $OUT/summary.txt | head -8
find $OUT -name "*.db" -delete
