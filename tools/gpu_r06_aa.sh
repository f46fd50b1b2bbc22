#!/bin/bash
# Round 6: the second log set inside the first one's buffers -- parity, then the 1e7 call after 1e8 calls again
OUT=gpurun_out/r06_aa; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_estimator_pipelines.py tests/test_full_size_configs.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
timeout 600 python tools/exp_switch_sizes.py > $OUT/switch.txt 2>&1; tail -n 24 $OUT/switch.txt
