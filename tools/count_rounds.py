"""Sweep rounds / event passes of the wave kernel (debug counters):  python tools/count_rounds.py n_packets key=value[,...] ..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic, _lib
from tardis_amd.engine import Engine
import ctypes as C

kw = dict(synthetic.BASELINE_CONFIGS[2])
kw["n_packets"] = int(sys.argv[1])
if os.environ.get("N_SHELLS"):
    kw["n_shells"] = int(os.environ["N_SHELLS"])
if os.environ.get("N_LINES"):
    kw["n_lines"] = int(os.environ["N_LINES"])
prob = synthetic.make_problem(seed=1, **kw)
for spec in sys.argv[2:]:
    out = {}
    for flag in (16, 32):
        eng = Engine(0)
        for kv in spec.split(","):
            k, v = kv.split("=")
            eng.set_option(k, int(v))
        eng.set_option("debug_flags", flag)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        for i in range(2):
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
        out[flag] = eng.last_counters()
        out["ms"] = eng.last_kernel_times()["propagate_ms"]
        eng.close()
    c = list(out[16].values()); out[32] = list(out[32].values())
    print(f"{spec:60s} prop {out['ms']:.2f} ms visits {c[0]:.3e} events {c[1]:.3e} rounds {c[7]:.3e} passes {out[32][7]:.3e}"
          f"  visits/round {c[0] / c[7]:.1f}  events/pass {c[1] / out[32][7]:.1f}", flush=True)
