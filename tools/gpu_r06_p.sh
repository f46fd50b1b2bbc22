#!/bin/bash
# Round 6: result streaming, stage by stage at the headline's packet count, + the rest of the streaming tests
OUT=gpurun_out/r06_p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_boundary_gpu.py -m gpu -x -q -k "stream" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 5 $OUT/pytest.log
STREAM=0 timeout 600 python tools/time_boundary.py 1e8 3 > $OUT/stages_plain.txt 2>&1
STREAM=1 timeout 600 python tools/time_boundary.py 1e8 3 > $OUT/stages_stream.txt 2>&1
tail -n 34 $OUT/stages_plain.txt; tail -n 36 $OUT/stages_stream.txt
