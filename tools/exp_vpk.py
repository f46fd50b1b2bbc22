"""Wave kernel vs group kernel on v-packet shapes (GPU box): python tools/exp_vpk.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import synthetic
from tardis_amd.engine import Engine
cases = [(20, 30_000, "downbranch", 10, 1_000_000), (20, 500_000, "macroatom", 10, 300_000), (100, 30_000, "downbranch", 10, 300_000),
         (100, 500_000, "macroatom", 10, 200_000), (100, 500_000, "macroatom", 3, 300_000), (50, 100_000, "macroatom", 10, 300_000)]
if len(sys.argv) > 1:
    cases = [cases[int(a)] for a in sys.argv[1:]]
for S, L, mode, nv, P in cases:
    prob = synthetic.make_problem(seed=1, n_packets=1, n_shells=S, n_lines=L, line_interaction_type=mode, n_vpackets=nv)
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
    out = []
    for variant in (2, 1):
        eng.set_option("variant", variant)
        best = 1e30
        for _ in range(2):
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
            best = min(best, eng.last_propagate_ms())
        c = eng.last_counters()
        out.append(f"variant {variant}: {best:9.1f} ms {P / best / 1e3:8.3f} Mpkt/s")
    print(f"S={S:3d} L={L:6d} {mode:10s} nv={nv:2d} P={P}: " + " | ".join(out) + f"  vpackets/packet {c['vpackets'] / P:.0f} vp visits/packet {c['vpacket_line_visits'] / P:.0f} events/packet {c['events'] / P:.0f}", flush=True)
    eng.close()
