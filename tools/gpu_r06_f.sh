#!/bin/bash
# Round 6, call 6: split launches (option epoch_split) -- parity on the multi-epoch tests, then A/B
OUT=gpurun_out/r06_f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_config3_shape.py tests/test_estimator_pipelines.py tests/test_full_size_configs.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="epoch_split=0 epoch_split=1 epoch_split=0 epoch_split=1 epoch_split=1,est_accumulate=2"
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/split_heavy_1e8.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 epoch_split=0 epoch_split=1 > $OUT/split_uniform_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 4e7 epoch_split=0 epoch_split=1 epoch_split=0 epoch_split=1 > $OUT/split_heavy_4e7.log 2>&1
tail -n 30 $OUT/*.log
