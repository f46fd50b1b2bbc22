#!/bin/bash
# Round 6, call 5: the lean no-stop proof of the interleaved-table kernels -- parity, A/B; write traffic of the shell-sorted log
OUT=gpurun_out/r06_e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_round6_options.py tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py tests/test_round5_options.py tests/test_estimator_pipelines.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="sweep_table=0 sweep_table=1 sweep_table=0 sweep_table=1 sweep_table=2"
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 $E > $OUT/lean_heavy_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 $E > $OUT/lean_heavy_2e7.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 sweep_table=0 sweep_table=1 sweep_table=2 > $OUT/lean_uniform_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 sweep_table=0 sweep_table=1 ls_waves_per_simd=3,sweep_table=0 ls_waves_per_simd=3,sweep_table=1 > $OUT/lean_heavy_1.25e7.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/lean_config2_1e7.log 2>&1
cd /tmp
for SLV in 0 1; do
  for c in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout -k 5 600 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/$OUT/pmc_sl${SLV}_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --packets 20000000 --cpu-sample 0 --boundary-packets 0 --no-extra --option log_by_shell=$SLV --option ls_waves_per_simd=4 > $GRAFT_REPO_ROOT/$OUT/pmc_sl${SLV}_$tag.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python - > $OUT/pmc_summary.txt 2>&1 <<'PY'
import csv, glob, collections
for T in (0, 1):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/r06_e/pmc_sl{T}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "propagate_wave" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f"log_by_shell={T}")
    for k in sorted(tot):
        print(f"  {k:40s} dispatches {n[k]:3d}  sum {tot[k]:.6e}  per dispatch {tot[k] / max(n[k], 1):.6e}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
tail -n 30 $OUT/pytest.log $OUT/lean_*.log $OUT/pmc_summary.txt
