#!/bin/bash
# Round 6: drain compaction -- parity tests, then the headline shape with and without, at 1e8 and 1e7 packets
OUT=gpurun_out/r06_r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_drain_compaction.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 12 $OUT/pytest.log
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 drain_compact=0 drain_compact=16 drain_compact=0 drain_compact=8 drain_compact=24 drain_split=1 > $OUT/ab_1e8.txt 2>&1; cat $OUT/ab_1e8.txt
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 drain_compact=0 drain_compact=16 drain_compact=0 drain_compact=8 drain_compact=16,ls_waves_per_simd=3 drain_compact=0,ls_waves_per_simd=3 > $OUT/ab_1e7.txt 2>&1; cat $OUT/ab_1e7.txt
