"""Time the v-packet configuration (config-2 shape + n_v v-packets per interaction) for engine option sets."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n_pk, n_v = int(sys.argv[1]), int(sys.argv[2])
kw = dict(synthetic.BASELINE_CONFIGS[2]); kw["n_packets"] = n_pk; kw["n_vpackets"] = n_v
prob = synthetic.make_problem(seed=1, **kw)
for spec in sys.argv[3:]:
    eng = Engine(0)
    for kv in spec.split(","):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
    best = 1e9
    for i in range(3):
        eng.reset_estimators(); eng.propagate(); eng.synchronize(); best = min(best, eng.last_propagate_ms())
    c = eng.last_counters()
    print(f"{spec:40s} {best:8.2f} ms -> {n_pk / best / 1e3:.2f} Mpkt/s; v-line visits {c['vpacket_line_visits']:.3e} traced {c['reserved']:.3e} "
          f"(x{c['reserved'] / max(c['vpacket_line_visits'], 1):.2f}), v-packets {c['vpackets']:.3e}", flush=True)
    eng.close()
