#!/bin/bash
# Round 6 final: the whole GPU suite + smoke with the final library, the rocprof recipe of the headline (kernel trace + PMC passes + traffic file), the driver's
# command line, configs[1]
OUT=gpurun_out/r06_final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
bash tools/gpu_profile_r06.sh r06_config3 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r06_config3/pmc_traffic_config3.json profiles/pmc_traffic_config3.json 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$?" >> $OUT/bench_driver_command.err
timeout 600 python bench.py --config 2 --steps 10 --warmup 3 > $OUT/bench_config2.json 2> $OUT/bench_config2.err
tail -n 6 $OUT/pytest.log $OUT/smoke.log; tail -n 25 gpurun_out/prof_r06_config3/summary.txt; tail -n 3 $OUT/bench_driver_command.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final/bench_driver_command.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','value_uniform_levels')}); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','kernel_ms','peak_measured','frac_of_measured')})
print('step', r['step']['estimator_passes_ms'], r['step']['device_ms'])
b=d['boundary']; print('boundary', b['ms'], b['device_ms'], b.get('full_size'), b['resident']['next_iteration_same_opacity']['ms'])
print('strong', d['strong_scaling_model']['device_ms'], d['strong_scaling_model']['efficiency_bound'])
print('extra', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['extra'].items()})
t=d['extra']['tardis_example_iteration']; print({k:(v.get('drop_in_call_ms'), v.get('device_ms'), v.get('longest_packet_events')) for k,v in t.items() if isinstance(v,dict)})
c=json.loads(open('gpurun_out/r06_final/bench_config2.json').read().strip().splitlines()[-1]); print('config2', c['value'], c['ms_per_step'], c['roofline']['frac'])
PY
