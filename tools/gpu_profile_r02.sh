#!/bin/bash
# Round-2 profiling recipe of the headline workload (run through gpurun): kernel trace + stats, then PMC passes, each in its
# own run (rocprofv3 --pmc must not be combined with the trace domains on this pool), and the FETCH_SIZE calibration.
#   tools/gpu_profile_r02.sh <tag> [bench args...]      e.g.  tools/gpu_profile_r02.sh r02_config3
set -u
TAG=${1:-r02_config3}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --boundary-packets 0 $*"
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.log" 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 600 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/calib" -o pmc --output-format csv -- python $ROOT/tools/micro_calib.py > "$OUT/calib.log" 2>&1
cd "$ROOT"
python tools/rocprof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python - "$OUT" <<'PY' >> "$OUT/summary.txt" 2>&1
import csv, glob, sys
print("== FETCH_SIZE calibration (tools/micro_calib.py: every lane reads random aligned blocks with 16-byte loads, 1.6 GB table)")
rows = []
for f in glob.glob(sys.argv[1] + "/calib/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "microbench" in r.get("Kernel_Name", "") and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
sizes = [16, 16, 32, 32, 64, 64, 128, 128]
for r, s in zip(rows, sizes):
    req = 4096 * 256 * 64 * s
    kib = float(r["Counter_Value"])
    print(f"dispatch {r['Dispatch_Id']}: blocks of {s:3d} B, requested {req / 1e9:7.3f} GB, FETCH_SIZE {kib:.0f} KiB = {kib * 1024 / 1e9:7.3f} GB, ratio FETCH_SIZE/requested {kib * 1024 / req:.3f}")
PY
grep -h '"metric"' "$OUT/trace_bench.log" > "$OUT/bench_line.json"
find "$OUT" -name "*.db" -delete
tail -12 "$OUT/summary.txt"
