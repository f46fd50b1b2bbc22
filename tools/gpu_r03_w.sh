#!/bin/bash
mkdir -p gpurun_out
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03w_bench_driver_cmd.json 2> gpurun_out/r03w.err
echo "wall seconds: $SECONDS"; tail -c 300 gpurun_out/r03w.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03w_bench_driver_cmd.json'))
print(d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'].get('per_packet_bit_exact'), d['boundary'].get('device_ms'))
for k,v in d.get('extra',{}).items():
    print(k, v.get('value',0)/1e6, v.get('ms_per_step'), v.get('error'), v.get('skipped'))
PY
