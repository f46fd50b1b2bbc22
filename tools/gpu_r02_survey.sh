#!/bin/bash
# Round-2 evidence run on the GPU box: memory-system ceilings for the access patterns of the macro-atom walk, then the
# kernel trace and PMC passes of the propagation kernel on the BASELINE configs[2] table shape.
#   tools/gpu_r02_survey.sh <tag> <packets> [extra bench args]
set -u
TAG=${1:-r02_cfg3}; PKTS=${2:-5000000}; shift 2 || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ "${MICRO:-0}" = "1" ]; then python tools/micro_r02.py > "$OUT/micro.txt" 2>&1; fi
BENCH="python $ROOT/bench.py --config 3 --packets $PKTS --steps 2 --warmup 1 --cpu-sample 0 --boundary-packets 0 $*"
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.log" 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
grep -h '"metric"' "$OUT/trace_bench.log" > "$OUT/bench_line.json"
find "$OUT" -name "*.db" -delete
tail -5 "$OUT/summary.txt"
