#!/bin/bash
# the driver's command line, nothing else
OUT=gpurun_out/r06_bench_final; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s.%N)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$? wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0) s" >> $OUT/bench_driver_command.err
tail -n 1 $OUT/bench_driver_command.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_final/bench_driver_command.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','value_uniform_levels')}); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','kernel_ms','launches_per_step','peak_measured','frac_of_measured')}); print(r['step'])
b=d['boundary']; print('boundary', b['ms'], b['device_ms'], {k:b['full_size'].get(k) for k in ('ms','device_ms','skipped')}, b['resident'])
print('strong', d['strong_scaling_model'])
print('extra', json.dumps(d['extra'])[:1500])
print('cpu', d['cpu_baseline'])
PY
