"""v-packet workloads: time per call and counters, with and without the prefix-sum screening (debug flag 33554432 switches it off).
   python tools/exp_vpk.py shape n_packets [spec ...]     shape: config5 (100 shells, 5e5 lines, macroatom, n_v 10) | config2v (20 shells,
   3e4 lines, downbranch, n_v 10)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine

shape = {"config5": dict(n_shells=100, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=10),
         "config2v": dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=10)}[sys.argv[1]]
P = int(float(sys.argv[2]))
specs = sys.argv[3:] or ["debug_flags=0", "debug_flags=33554432"]
prob = synthetic.make_problem(seed=1, n_packets=1, **shape)
eng = Engine(0)
eng.set_option("track_last_interaction", 0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
ref = None
for spec in specs:
    for kv in spec.split(","):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    best = 1e30
    for rep in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        best = min(best, eng.last_propagate_ms())
    c = eng.last_counters()
    r = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    sig = (c["line_visits"], c["events"], c["vpackets"], c["vpacket_line_visits"], c["rng_draws"], float(r.v_packets_energy_hist.sum()))
    if ref is None:
        ref = (sig, r.output_nus.copy(), r.v_packets_energy_hist.copy())
    same = sig[:5] == ref[0][:5] and (r.output_nus == ref[1]).all()
    import numpy as np
    dh = float(np.max(np.abs(r.v_packets_energy_hist - ref[2])) / max(np.max(np.abs(ref[2])), 1e-300))
    print(f"{spec:36s} variant {eng.last_variant()}  {best:9.1f} ms  {P / best / 1e3:7.3f} Mpkt/s  v-packets/pkt {c['vpackets'] / P:.0f}  v-visits/pkt {c['vpacket_line_visits'] / P:.0f}  "
          f"traced/committed {c['reserved'] / max(c['vpacket_line_visits'], 1):.2f}  {'same' if same else 'DIFFERENT'}  vhist max rel diff {dh:.1e}", flush=True)
    for kv in spec.split(","):
        k, v = kv.split("=")
        eng.set_option(k, 0 if k == "debug_flags" else -1 if k == "variant" else int(v))
eng.close()
