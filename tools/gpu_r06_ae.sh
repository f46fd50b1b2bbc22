#!/bin/bash
# Round 6: issue priority for waves that carry a long-lived packet (debug_flags 1 << 30) -- A/B at 1.25e7 and 1e8 packets
OUT=gpurun_out/r06_ae; mkdir -p $OUT; export TMPDIR=/tmp
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 debug_flags=0 debug_flags=1073741824 debug_flags=0 debug_flags=1073741824 > $OUT/ab_1e7.txt 2>&1; cat $OUT/ab_1e7.txt | cut -c1-200
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 debug_flags=0 debug_flags=1073741824 debug_flags=0 debug_flags=1073741824 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt | cut -c1-200
