#!/bin/bash
# record lines of the other BASELINE configurations for profiles/ (round 3)
mkdir -p gpurun_out
timeout 600 python bench.py --config 2 --steps 10 --warmup 2 --no-extra > gpurun_out/r03_bench_config2.json 2> gpurun_out/r03u.err
timeout 600 python bench.py --config 2 --vpackets 10 --packets 2000000 --steps 3 --warmup 1 --boundary-packets 0 --no-extra > gpurun_out/r03_bench_config2_vpackets10_2e6pkts.json 2>> gpurun_out/r03u.err
timeout 900 python bench.py --config 5 --steps 1 --warmup 1 --cpu-sample 3000 --boundary-packets 0 --no-extra > gpurun_out/r03_bench_config5_full_6.25e7pkts.json 2>> gpurun_out/r03u.err
tail -c 300 gpurun_out/r03u.err
python - <<'PY'
import json
for f in ("r03_bench_config2", "r03_bench_config2_vpackets10_2e6pkts", "r03_bench_config5_full_6.25e7pkts"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:40], d.get("cpu_baseline", {}).get("per_packet_bit_exact"))
    except Exception as e:
        print(f, "FAILED", e)
PY
