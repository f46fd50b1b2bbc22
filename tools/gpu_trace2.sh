#!/bin/bash
# kernel trace + stats of the secondary workloads in one call (bounded):  tools/gpu_trace2.sh
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace2; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d "$OUT/c3/trace" -o trace -- python $ROOT/bench.py --config 3 --packets 5000000 --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/c3.log" 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d "$OUT/vp/trace" -o trace -- python $ROOT/bench.py --config 2 --packets 2000000 --vpackets 10 --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/vp.log" 2>&1
cd "$ROOT"
for t in c3 vp; do python tools/rocprof_summary.py "$OUT/$t" > "$OUT/$t.summary.txt" 2>&1; grep '"metric"' "$OUT/$t.log" | cut -c1-250 >> "$OUT/$t.summary.txt"; done
find "$OUT" -name "*.db" -delete
head -8 "$OUT/c3.summary.txt" | cut -c1-140; head -8 "$OUT/vp.summary.txt" | cut -c1-140
