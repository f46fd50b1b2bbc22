#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/exp_drain.py 1e6 5e6 1e7 2e7 4e7 > gpurun_out/r03c_drain.txt 2>&1
cat gpurun_out/r03c_drain.txt
timeout 600 python -m pytest tests/test_multirank_gpu.py tests/test_next_rows_golden.py tests/test_packet_source.py tests/test_radfield.py tests/test_volley_queue.py -m gpu -x -q 2>&1 | tail -3
