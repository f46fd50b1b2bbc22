#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03i_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r03i_tests.log | tail -2
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err
tail -c 400 gpurun_out/r03i_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03i_bench.json'))
print(d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['launches_per_step'])
b=d.get('boundary',{})
print('boundary', b.get('ms'), b.get('device_ms'), {k:(v['ms'],v['device_ms']) for k,v in b.get('resident',{}).items() if isinstance(v,dict)})
for k,v in d.get('extra',{}).items():
    print(k, v['value']/1e6, v['ms_per_step'], v['setup_s'], v['roofline']['frac'], v.get('cpu_sample',{}).get('per_packet_bit_exact'))
PY
