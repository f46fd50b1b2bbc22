#!/bin/bash
# round 3, second GPU call: full GPU test suite, cache-policy A/B, section timers, default bench line with the new legs
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_tests.log 2>&1
tail -4 gpurun_out/r03b_tests.log
timeout 900 python tools/exp_policy.py 2e7 debug_flags=0 debug_flags=65536 debug_flags=131072 debug_flags=262144 debug_flags=524288 debug_flags=1048576 debug_flags=0 debug_flags=1900544 debug_flags=1966080 > gpurun_out/r03b_policy.txt 2>&1
cat gpurun_out/r03b_policy.txt
TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_timers.so timeout 600 python tools/sections_cfg3.py 2e7 debug_flags=0 > gpurun_out/r03b_sections.txt 2>&1
cat gpurun_out/r03b_sections.txt
timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
tail -c 600 gpurun_out/r03b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03b_bench.json'))
print(d['value']/1e6, d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('boundary'), indent=1)[:1500])
for k,v in d.get('extra',{}).items():
    print(k, v['value']/1e6, v['ms_per_step'], v['setup_s'], v['roofline']['frac'], v.get('cpu_sample',{}).get('per_packet_bit_exact'), v['roofline']['per_packet'])
PY
