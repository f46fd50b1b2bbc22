"""A/B of engine options / debug flags on the configs[2] table shape (one process, one box): propagate time per spec.
   python tools/exp_policy.py n_packets [heavy] spec spec ...     spec = key=value[,key=value...]  (debug_flags=N, variant=N, ...)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine

P = int(float(sys.argv[1]))
specs = sys.argv[2:]
level = "uniform"
if specs and specs[0] == "heavy":
    level, specs = "heavy", specs[1:]
shape = dict(n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
if os.environ.get("EXP_SHAPE") == "config2":
    shape = dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes=level, **shape)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
ref = None
for spec in specs:
    keys = []
    for kv in spec.split(","):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
        keys.append(k)
        if k == "walk_sector_packing":  # (a table-layout option: takes effect at set_opacity)
            eng.set_opacity(prob.opacity_state)
    best = total = 1e30
    for rep in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        kt = eng.last_kernel_times()
        best = min(best, kt["propagate_ms"])
        total = min(total, eng.last_propagate_ms())
    c = eng.last_counters()
    sig = (c["line_visits"], c["events"], c["macro_transitions"], c["rng_draws"])
    if ref is None:
        ref = sig
    print(f"{spec:40s} step {total:9.2f} ms = {P / total / 1e3:6.2f} Mpkt/s;  propagate {best:9.2f} ms  {P / best / 1e3:7.2f} Mpkt/s  launches {kt['launches']}  est {kt['estimator_ms']:.1f} ms  "
          f"{'same' if sig == ref else 'COUNTERS DIFFER'}", flush=True)
    for k in keys:  # back to defaults
        eng.set_option(k, {"debug_flags": 0, "variant": -1}.get(k, 0)) if k in ("debug_flags", "variant") else None
eng.close()
