#!/bin/bash
# Round 6, call 3: the shell-sorted log (option log_by_shell) -- parity, then A/B; the boundary's stages; the streaming-rate kernel
OUT=gpurun_out/r06_c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_estimator_pipelines.py tests/test_round6_options.py tests/test_round5_options.py tests/test_hip_parity.py tests/test_config3_shape.py tests/test_heavy_blocks.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
E="log_sets=1,log_by_shell=0 log_sets=1,log_by_shell=1 log_sets=1,log_by_shell=0 log_sets=1,log_by_shell=1 log_by_shell=0 log_by_shell=1"
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 2e7 $E > $OUT/sl_heavy_2e7.log 2>&1
EXP_LEVELS=heavy timeout 900 python tools/exp_cfg3.py 1e8 log_by_shell=0 log_by_shell=1 log_by_shell=0 log_by_shell=1 log_by_shell=1,est_accumulate=2 > $OUT/sl_heavy_1e8.log 2>&1
timeout 900 python tools/exp_cfg3.py 1e8 log_by_shell=0 log_by_shell=1 > $OUT/sl_uniform_1e8.log 2>&1
EXP_LEVELS=heavy timeout 600 python tools/exp_cfg3.py 1.25e7 log_by_shell=0,ls_waves_per_simd=3 log_by_shell=1,ls_waves_per_simd=3 log_by_shell=0 log_by_shell=1 > $OUT/sl_heavy_1.25e7.log 2>&1
EXP_SHAPE=config2 timeout 600 python tools/exp_cfg3.py 1e7 $E > $OUT/sl_config2_1e7.log 2>&1
timeout 600 python tools/time_boundary.py 1e7 3 > $OUT/boundary_config3.log 2>&1
timeout 600 python tools/time_boundary.py 1e7 2 > $OUT/boundary_config2.log 2>&1
python - > $OUT/stream_peak.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from tardis_amd.engine import Engine
eng = Engine(0)
for n, blocks in ((1 << 28, 1024), (1 << 28, 2048), (1 << 28, 4096), (1 << 28, 16384), (1 << 29, 4096)):
    ms = eng.debug_microbench(15, n, 4, blocks)
    print(f"copy n_doubles={n} blocks={blocks}: {ms:.3f} ms -> {n * 8 * 4 / (ms * 1e-3) / 1e12:.3f} TB/s")
eng.close()
PY
tail -n 40 $OUT/*.log
