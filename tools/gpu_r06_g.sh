#!/bin/bash
# Round 6, call 7: the whole GPU suite, the default bench line (short), cut-off sweep with the lean sweep
OUT=gpurun_out/r06_g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" >> $OUT/bench_default.err
E="lane_sweep_min_active=12,walk_min_active=12 lane_sweep_min_active=8,walk_min_active=12 lane_sweep_min_active=16,walk_min_active=12 lane_sweep_min_active=20,walk_min_active=12 lane_sweep_min_active=12,walk_min_active=8 lane_sweep_min_active=12,walk_min_active=16 lane_sweep_min_active=16,walk_min_active=16 lane_sweep_min_active=12,walk_min_active=12"
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/cutoffs_heavy_1e8.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
tail -n 15 $OUT/pytest.log $OUT/bench_default.err $OUT/cutoffs_heavy_1e8.log $OUT/smoke.log; python -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','rccl_ranks')}); print(d['roofline']['frac'], d['roofline'].get('peak_measured'), d['roofline'].get('frac_of_measured'), d['roofline']['kernel_ms'])
print('boundary', {k:d['boundary'][k] for k in ('ms','device_ms')}, d['boundary']['resident']['next_iteration_same_opacity'])
print('strong', d.get('strong_scaling_model'))
print('extra', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['extra'].items()})
print(json.dumps(d['extra'].get('tardis_example_iteration'))[:1500])
"
