#!/bin/bash
# quick PMC passes for one bench configuration:  tools/gpu_pmc_quick.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- bench args
set -u
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 $*"
cd /tmp
i=0
for c in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 5 60 rocprofv3 --pmc $c -d "$OUT/pmc_$i" -o pmc --output-format csv -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "propagate_wave" not in k: continue
        key = r["Counter_Name"]; acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
nd = None
for k, (v, n) in sorted(acc.items()):
    print(f"{k:40s} per_dispatch={v / max(n,1) * 1.0:.6e}  (rows {n})")
PY
find "$OUT" -name "*.db" -delete
