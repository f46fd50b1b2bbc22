#!/bin/bash
# Round 6: one partition pass where the bins are few (configs[1] / tardis_example shapes) -- parity, then configs[1] with and without
OUT=gpurun_out/r06_ab; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_estimator_pipelines.py tests/test_hip_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 est_one_level=1 est_one_level=0 est_one_level=1 est_one_level=0 > $OUT/ab_cfg2.txt 2>&1; cat $OUT/ab_cfg2.txt
