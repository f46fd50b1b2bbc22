#!/bin/bash
# configs[4] shape, group kernel: PMC passes (each in its own run, no trace domains) -> gpurun_out/cfg5pmc/summary.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/cfg5pmc; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  EXP_SHAPE=config5 timeout -k 5 300 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- python $ROOT/tools/exp_cfg3.py 1e6 variant=1 > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -h "variant=1" "$OUT"/pmc_1.log | cut -c1-200
grep "propagate_group" "$OUT/summary.txt" | cut -c1-200
