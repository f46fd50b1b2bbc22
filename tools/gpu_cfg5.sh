#!/bin/bash
# configs[4] shape (100 shells, 5e5 lines, macroatom, 10 v-packets) at 3e6 packets: bench line + kernel trace (GPU box)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cfg5; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $ROOT/bench.py --config 5 --packets 3000000 --steps 2 --warmup 1 --boundary-packets 0 > "$OUT/bench.log" 2>&1
cd "$ROOT"
grep -h '"metric"' "$OUT/bench.log" > "$OUT/bench_line.json"
python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
head -12 "$OUT/summary.txt" | cut -c1-170
cut -c1-400 "$OUT/bench_line.json"
