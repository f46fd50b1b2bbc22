#!/bin/bash
# A few PMC passes of the propagation kernel for one bench configuration:  tools/gpu_pmc_lite.sh <tag> <bench args...>
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --boundary-packets 0 $*"
cd /tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -h '"metric"' "$OUT/pmc_1.log" > "$OUT/bench_line.json"
find "$OUT" -name "*.db" -delete
grep "propagate_wave" "$OUT/summary.txt" | cut -c62-200
