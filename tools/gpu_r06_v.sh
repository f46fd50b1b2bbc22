#!/bin/bash
# Round 6: log_sets automatic (one set + longer epochs for calls of several epochs) -- parity, then A/B at 1e8 / 4e7 / 2e7 packets
OUT=gpurun_out/r06_v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_estimator_pipelines.py tests/test_drain_compaction.py tests/test_boundary_gpu.py tests/test_full_size_configs.py tests/test_round6_options.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 6 $OUT/pytest.log
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 log_sets=2 log_sets=0 log_sets=2 log_sets=0 log_sets=1,log_capacity=2160000000 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 4e7 log_sets=2 log_sets=0 log_sets=2 log_sets=0 > $OUT/ab_4e7.txt 2>&1; cat $OUT/ab_4e7.txt
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 log_sets=2 log_sets=0 log_sets=1 > $OUT/ab_2e7.txt 2>&1; cat $OUT/ab_2e7.txt
