#!/bin/bash
# Round-end evidence in one gpurun call: default bench line, kernel trace + stats, HBM traffic PMC passes (each bounded).
#   tools/gpu_final.sh <tag>   -> gpurun_out/final_<tag>/{bench_default.json,summary.txt,pmc_traffic.json}
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python bench.py > "$OUT/bench_default.log" 2>&1
grep '"metric"' "$OUT/bench_default.log" | tail -1 > "$OUT/bench_default.json"
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0"
cd /tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.log" 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
grep '"metric"' "$OUT/trace_bench.log" | cut -c1-300 >> "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
tail -3 "$OUT/summary.txt" | cut -c1-200
cat "$OUT/pmc_traffic.json"
