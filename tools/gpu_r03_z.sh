#!/bin/bash
# kernel traces of the two extra workloads of the bench line (round 3)
set -u
ROOT=$(pwd); export TMPDIR=/tmp
for tag in config5_shape_1e7pkts heavy_tail; do
  OUT=$ROOT/gpurun_out/trace_$tag; mkdir -p "$OUT"
  if [ $tag = heavy_tail ]; then ARGS="--level-sizes heavy --steps 1 --warmup 1"; else ARGS="--config 5 --packets 10000000 --steps 1 --warmup 1"; fi
  cd /tmp
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python $ROOT/bench.py $ARGS --cpu-sample 0 --boundary-packets 0 --no-extra > "$OUT/bench.log" 2>&1
  cd "$ROOT"
  python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
  grep -h '"metric"' "$OUT/bench.log" > "$OUT/bench_line.json"
  find "$OUT" -name "*.db" -delete
  head -8 "$OUT/summary.txt" | cut -c1-150
done
