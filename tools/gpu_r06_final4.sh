#!/bin/bash
# Round 6, last call: the whole GPU suite + smoke with the final library
OUT=gpurun_out/r06_final6; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
grep -E "passed|failed|rc=" $OUT/pytest.log | tail -3; tail -n 2 $OUT/smoke.log
