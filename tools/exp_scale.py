"""Kernel time vs packets per launch on the configs[2] table shape (GPU box): separates the drain of a launch from its
throughput.  python tools/exp_scale.py "opts" P1 P2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

opts = sys.argv[1]
sizes = [int(float(a)) for a in sys.argv[2:]]
shape = dict(n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
prob = synthetic.make_problem(seed=1, n_packets=1, **shape)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
for kv in opts.split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
for P in sizes:
    eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
    best = 1e30
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        best = min(best, eng.last_kernel_times()["propagate_ms"])
    c = eng.last_counters()
    print(f"{opts:40s} P={P:>9d}  propagate {best:9.2f} ms  {P / best / 1e3:7.2f} Mpkt/s  events/packet {c['events'] / P:.1f}", flush=True)
eng.close()
