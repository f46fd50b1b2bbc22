#!/bin/bash
# A/B of the sector-aware packing of the compact walk tables: time in one process, L2 -> fabric requests under rocprofv3 --pmc
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03k; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python tools/exp_policy.py 2e7 walk_sector_packing=1 walk_sector_packing=0 walk_sector_packing=1 walk_sector_packing=0 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
cd /tmp
for pk in 1 0; do
  timeout -k 5 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_pack$pk" -o pmc -- python $ROOT/tools/exp_policy.py 2e7 walk_sector_packing=$pk > "$OUT/pmc_pack$pk.log" 2>&1
done
cd "$ROOT"
for pk in 1 0; do
  mkdir -p $OUT/s$pk; rm -rf $OUT/s$pk/pmc_1; mv $OUT/pmc_pack$pk $OUT/s$pk/pmc_1
  echo "== walk_sector_packing=$pk" >> $OUT/summary.txt
  python tools/rocprof_summary.py $OUT/s$pk | grep propagate_wave >> $OUT/summary.txt
done
find "$OUT" -name "*.db" -delete
cat $OUT/summary.txt | cut -c1-200
