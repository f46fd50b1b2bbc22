#!/bin/bash
# Kernel trace only (no PMC passes):  tools/gpu_trace.sh <tag> [bench args...]  -> gpurun_out/trace_<tag>/summary.txt
set -u
TAG=${1:-t}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 "$@" > "$OUT/bench.log" 2>&1
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep '"metric"' "$OUT/bench.log" | cut -c1-200 >> "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
