"""Round 6 probe: what does the propagation kernel at twelve waves per CU lose when a streaming kernel with small workgroups runs beside it on the free wave slots?
(The question behind co-scheduling the estimator passes: are they -- ~250 GB of streaming traffic per epoch -- hideable beside a propagation launch at all?)
    python tools/probe_coresidency.py [packets] [copy blocks]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prob = synthetic.make_problem(seed=1, n_packets=1, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", level_sizes="heavy")
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
side = Engine(0)
N = 1 << 27  # 1 GiB table: 1 GiB of traffic per pass


def run(wps, with_copy):
    eng.set_option("ls_waves_per_simd", wps)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()  # warm
    stop = threading.Event()
    moved = [0.0, 0.0]

    def copier():
        t0 = time.perf_counter()
        while not stop.is_set():
            side.debug_microbench(15, N, 8, blocks)  # (two launches of 8 passes each; the second is the timed one)
            moved[0] += 2 * 8 * N * 8
        moved[1] = time.perf_counter() - t0
    th = threading.Thread(target=copier)
    if with_copy:
        th.start()
        time.sleep(0.05)
    t0 = time.perf_counter()
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    if with_copy:
        th.join()
    kt = eng.last_kernel_times()
    print(f"waves/SIMD {wps}  copy {'on ' if with_copy else 'off'} (blocks {blocks}): call {1e3 * dt:8.1f} ms  propagate {kt['propagate_ms']:8.1f} ms x{kt['launches']}  "
          f"est {kt['estimator_ms']:7.1f} ms" + (f"   copied {moved[0] / 1e9:7.1f} GB in {moved[1]:.3f} s = {moved[0] / moved[1] / 1e12:.2f} TB/s" if with_copy else ""), flush=True)


for wps in (4, 3):
    for with_copy in (False, True, False, True):
        run(wps, with_copy)
eng.close(); side.close()
