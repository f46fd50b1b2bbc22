#!/bin/bash
# Round 6, call 2: new tests (one-shot run, sweep table, estimator pipelines incl. the dyadic accumulate), A/B of the accumulate kernels, the counters of the
# sweep-table A/B (TCP_TCC_READ_REQ, TCP_PENDING_STALL), the streaming-peak microbench.
OUT=gpurun_out/r06_b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_one_shot_run.py tests/test_round6_options.py tests/test_estimator_pipelines.py tests/test_round5_options.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="log_sets=1,est_accumulate=1 log_sets=1,est_accumulate=2 log_sets=1,est_accumulate=1 log_sets=1,est_accumulate=2 est_accumulate=1 est_accumulate=2"
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 $E > $OUT/acc_heavy_2e7.log 2>&1
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 est_accumulate=1 est_accumulate=2 est_accumulate=1 est_accumulate=2 > $OUT/acc_heavy_1e8.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/acc_config2_1e7.log 2>&1
python - > $OUT/stream_peak.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from tardis_amd.engine import Engine
eng = Engine(0)
for n, blocks in ((1 << 27, 4096), (1 << 28, 4096), (1 << 28, 2048), (1 << 28, 8192), (1 << 29, 4096)):
    ms = eng.debug_microbench(15, n, 4, blocks)
    print(f"copy n_doubles={n} blocks={blocks}: {ms:.3f} ms -> {n * 8 * 4 / (ms * 1e-3) / 1e12:.3f} TB/s")
eng.close()
PY
cd /tmp
for T in 0 1; do
  for c in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout -k 5 600 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/$OUT/pmc_t${T}_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --packets 40000000 --cpu-sample 0 --boundary-packets 0 --no-extra --option sweep_table=$T --option ls_waves_per_simd=4 > $GRAFT_REPO_ROOT/$OUT/pmc_t${T}_$tag.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - > $OUT/pmc_summary.txt 2>&1 <<'PY'
import csv, glob, collections
for T in (0, 1):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/r06_b/pmc_t{T}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "propagate_wave" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f"sweep_table={T}")
    for k in sorted(tot):
        print(f"  {k:40s} dispatches {n[k]:3d}  sum {tot[k]:.6e}  per dispatch {tot[k] / max(n[k], 1):.6e}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
tail -n 25 $OUT/pytest.log $OUT/acc_*.log $OUT/stream_peak.log $OUT/pmc_summary.txt
