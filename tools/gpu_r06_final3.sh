#!/bin/bash
# Round 6: the whole GPU suite, smoke, the driver command line (after: streaming, drain_compact option, accumulate LOOP default)
OUT=gpurun_out/r06_final5; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
T0=$(date +%s.%N)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$? wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0) s" >> $OUT/bench_driver_command.err
grep -E "passed|failed" $OUT/pytest.log | tail -2; tail -n 2 $OUT/smoke.log; tail -n 1 $OUT/bench_driver_command.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final5/bench_driver_command.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','value_uniform_levels')}); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','kernel_ms','peak_measured','frac_of_measured')})
b=d['boundary']; print('boundary', b['ms'], b['device_ms'], {k:b['full_size'].get(k) for k in ('ms','device_ms','skipped')}, b['resident']['next_iteration_same_opacity']['ms'])
print('strong', d['strong_scaling_model']['device_ms'], d['strong_scaling_model']['efficiency_bound'])
print('extra', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['extra'].items()})
PY
