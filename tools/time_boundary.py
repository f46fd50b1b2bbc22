"""Where the host / PCIe time of one drop-in call goes (GPU box): the stages of transport.montecarlo_transport_with_vpackets timed one by one
on host arrays.   python tools/time_boundary.py [packets] [config: 2 | 3]   (STREAM=1: with result streaming)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import state as st, synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
cfg_no = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = dict(synthetic.BASELINE_CONFIGS[cfg_no]); kw.pop("n_packets")
if os.environ.get("BOUNDARY_MODE"):  # e.g. macroatom on the configs[0] tables: a tardis_example iteration
    kw["line_interaction_type"] = os.environ["BOUNDARY_MODE"]
prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes="heavy" if kw["line_interaction_type"] == "macroatom" else "uniform", **kw)
pc = synthetic.black_body_packets(n, float(prob.geometry.r_inner[0]), 1.0e4)
eng = Engine(0)
STREAM = bool(int(os.environ.get("STREAM", "0")))


def t(label, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
    print(f"  {label:34s} {1e3 * dt:9.2f} ms", flush=True)
    return r


for rep in range(3):
    print(f"--- call {rep} ({n} packets, config {cfg_no})")
    t0 = time.perf_counter()
    t("set_geometry", eng.set_geometry, prob.geometry, prob.time_explosion)
    t("set_opacity", eng.set_opacity, prob.opacity_state)
    t("set_config", eng.set_config, prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_option("track_last_interaction", 1)
    t("set_packets", eng.set_packets, pc)
    t("reset_estimators", eng.reset_estimators)
    if STREAM:  # the caller's arrays registered before the call (what transport.montecarlo_transport_with_vpackets does): filled launch by launch
        trk = t("allocate trackers (host)", st.LastInteractionTrackers, n)
        t("stream_results", eng.stream_results, pc.output_nus, pc.output_energies, trk)
    t("propagate + synchronize", lambda: (eng.propagate(), eng.synchronize()))
    print("  kernel times", eng.last_kernel_times(), " estimator passes", eng.last_estimator_ms() if hasattr(eng, "last_estimator_ms") else None)
    if STREAM:
        print("  streamed / resent packets", eng.streamed_packets(), " launches", eng.last_kernel_times()["launches"])
    else:
        trk = t("allocate trackers (host)", st.LastInteractionTrackers, n)
    res = t("get_results (all)", eng.get_results, pc.output_nus, pc.output_energies, True, trackers=trk)
    print(f"  {'total':34s} {1e3 * (time.perf_counter() - t0):9.2f} ms   device {eng.last_propagate_ms():.2f} ms")
    t("get_results (no trackers)", eng.get_results, pc.output_nus, pc.output_energies, False)
    t("get_results (no trk, no line est)", eng.get_results, pc.output_nus, pc.output_energies, False, False)
    t("get_results (nothing per packet)", eng.get_results, None, None, False, False, None, False)
eng.close()
