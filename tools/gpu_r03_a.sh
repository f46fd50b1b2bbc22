#!/bin/bash
# round 3, first GPU call: heavy-tailed macro-atom blocks -- parity tests, then the rate next to the uniform workload
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_heavy_blocks.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r03a_tests.log 2>&1
tail -5 gpurun_out/r03a_tests.log
timeout 600 python bench.py --level-sizes heavy --packets 20000000 --steps 2 --warmup 1 --cpu-sample 20000 --boundary-packets 0 > gpurun_out/r03a_heavy_2e7.json 2> gpurun_out/r03a_heavy_2e7.err
tail -c 1500 gpurun_out/r03a_heavy_2e7.json
timeout 600 python bench.py --packets 20000000 --steps 2 --warmup 1 --cpu-sample 0 --boundary-packets 0 > gpurun_out/r03a_uniform_2e7.json 2> gpurun_out/r03a_uniform_2e7.err
tail -c 800 gpurun_out/r03a_uniform_2e7.json
