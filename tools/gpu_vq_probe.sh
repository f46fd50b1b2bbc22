#!/bin/bash
# volley-queue probe (GPU box): kernel split of a variant-4 call on the configs[4] shape (the tracer's first launches are full)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/vqprobe; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
EXP_SHAPE=config5 timeout -k 5 400 rocprofv3 --kernel-trace -d "$OUT/trace" -o trace -- python $ROOT/tools/exp_cfg3.py 1e6 variant=4 > "$OUT/trace.log" 2>&1
cd "$ROOT"
python - "$OUT" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/trace/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
a = [(e - s) / 1e3 for n, s, e in rows if "propagate_wave" in n]
b = [(e - s) / 1e3 for n, s, e in rows if "vpacket_trace" in n]
print("launches", len(a), len(b))
for k in (0, 1, 2, 3, 5, 10, 20, 50, 100, 150, 200, 300, 400, 600, 1000, 2000, 5000, 10000, 20000):
    if k < len(b): print(f"epoch {k:6d}: propagate {a[k]:10.1f} us   tracer {b[k]:10.1f} us")
import itertools
print("sum propagate ms", sum(a) / 1e3, "sum tracer ms", sum(b) / 1e3)
half = len(a) // 2
print("first run: propagate", sum(a[:half]) / 1e3, "tracer", sum(b[:half]) / 1e3)
acc = 0.0
for k in range(half):
    acc += b[k]
    if k in (10, 50, 100, 200, 400, 800, 1600): print(f"tracer cumulative after {k} epochs: {acc / 1e3:.1f} ms, propagate {sum(a[:k + 1]) / 1e3:.1f} ms")
PY
find "$OUT" -name "*.db" -delete
tail -2 "$OUT/trace.log" | cut -c1-200
