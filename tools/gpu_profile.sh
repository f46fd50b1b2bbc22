#!/bin/bash
# GPU-box profiling recipe (run through gpurun).  Kernel trace + stats, then PMC passes in their own runs.
#   tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1
done
cd "$ROOT"
find "$OUT" -name "*.csv" | head -50
