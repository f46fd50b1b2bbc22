"""A 1e7-packet drop-in call after 1e8-packet calls on the same context (the order of bench.py's legs): where does the time of the smaller call go?
    python tools/exp_switch_sizes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import state as st, synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

kw = dict(synthetic.BASELINE_CONFIGS[3]); kw.pop("n_packets")
prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes="heavy", **kw)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)


def t(label, f, *a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
    print(f"  {label:34s} {1e3 * dt:9.2f} ms", flush=True)
    return r


for big in range(2):
    eng.create_blackbody_packets(100_000_000, float(prob.geometry.r_inner[0]), 1.0e4)
    t("1e8 resident: propagate", lambda: (eng.reset_estimators(), eng.propagate(), eng.synchronize()))
    print("   launches", eng.last_kernel_times()["launches"])
pc = synthetic.black_body_packets(10_000_000, float(prob.geometry.r_inner[0]), 1.0e4)
trk = st.LastInteractionTrackers(10_000_000)
for rep in range(3):
    print("--- 1e7 host-array call", rep)
    t("set_opacity", eng.set_opacity, prob.opacity_state)
    t("set_packets", eng.set_packets, pc)
    t("propagate + synchronize", lambda: (eng.reset_estimators(), eng.propagate(), eng.synchronize()))
    print("   launches", eng.last_kernel_times()["launches"], "device", eng.last_propagate_ms())
    t("get_results", eng.get_results, pc.output_nus, pc.output_energies, True, trackers=trk)
eng.create_blackbody_packets(100_000_000, float(prob.geometry.r_inner[0]), 1.0e4)
t("1e8 resident again: propagate", lambda: (eng.reset_estimators(), eng.propagate(), eng.synchronize()))
eng.close()
