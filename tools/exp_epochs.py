"""Epoch behaviour on the configs[2] shape: python tools/exp_epochs.py P "opts" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import synthetic
from tardis_amd.engine import Engine
P = int(float(sys.argv[1]))
prob = synthetic.make_problem(seed=1, n_packets=1, n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
for spec in sys.argv[2:]:
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
    for rep in range(2):
        t0 = time.perf_counter()
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        dt = time.perf_counter() - t0
    kt = eng.last_kernel_times()
    print(f"{spec:50s} wall {dt * 1e3:9.1f} ms  {P / dt / 1e6:6.2f} Mpkt/s  device {eng.last_propagate_ms():9.1f}  launches {kt['launches']}  "
          f"prop sum {kt['propagate_ms']:9.1f}  est sum {kt['estimator_ms']:9.1f}  prep {kt['seed_ms']:.1f}", flush=True)
eng.close()
