#!/bin/bash
# Round 6: passes beside the next launch (log_sets 2) or before it (log_sets 1), with the faster accumulate kernel
OUT=gpurun_out/r06_u; mkdir -p $OUT; export TMPDIR=/tmp
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 log_sets=2 log_sets=1 log_sets=2 log_sets=1 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt
