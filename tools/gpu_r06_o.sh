#!/bin/bash
# Round 6: result streaming -- the boundary tests, then the driver's command line (boundary and boundary.full_size legs)
OUT=gpurun_out/r06_o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_boundary_gpu.py tests/test_one_shot_run.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 15 $OUT/pytest.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_o/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}); r=d['roofline']; print({k:r[k] for k in ('frac','kernel_ms')})
b=d['boundary']; print('boundary', b['ms'], b['device_ms'], b['full_size'])
PY
