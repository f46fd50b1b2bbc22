#!/bin/bash
# Round 6: issue priority for waves that carry a long-lived packet -- threshold sweep at 1.25e7 and 1e8 packets
OUT=gpurun_out/r06_af; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round6_options.py -m gpu -x -q -k "priority" > $OUT/pytest.log 2>&1; tail -n 2 $OUT/pytest.log
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1.25e7 priority_events=0 priority_events=128 priority_events=512 priority_events=1024 priority_events=4096 priority_events=0 priority_events=256 priority_events=2048 priority_events=0 > $OUT/ab_1e7.txt 2>&1; cat $OUT/ab_1e7.txt | cut -c1-160
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 priority_events=0 priority_events=256 priority_events=1024 priority_events=0 priority_events=256 priority_events=1024 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt | cut -c1-160
