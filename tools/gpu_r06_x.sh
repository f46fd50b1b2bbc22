#!/bin/bash
# Round 6: drain compaction at a packing density below 64 lanes per wave (does the drain's chain stay short when the packed waves stay sparse?)
OUT=gpurun_out/r06_x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_drain_compaction.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -n 3 $OUT/pytest.log
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1.25e7 log_sets=2 log_sets=2,drain_compact=2 log_sets=2,drain_compact=2,drain_pack_lanes=8 log_sets=2,drain_compact=2,drain_pack_lanes=4 log_sets=2,drain_compact=4,drain_pack_lanes=8 log_sets=2,drain_compact=4,drain_pack_lanes=16 log_sets=2,drain_compact=1,drain_pack_lanes=2 log_sets=2 > $OUT/ab_1e7.txt 2>&1; cat $OUT/ab_1e7.txt
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 log_sets=2 log_sets=2,drain_compact=2,drain_pack_lanes=8 log_sets=2,drain_compact=4,drain_pack_lanes=16 log_sets=2 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt
