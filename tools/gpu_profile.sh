#!/bin/bash
# GPU-box profiling recipe (run through gpurun).  Kernel trace + stats first, then PMC passes, each in its own run
# (rocprofv3 --pmc must not be combined with the trace domains on this pool).
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
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC" \
         "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
