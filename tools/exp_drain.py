"""Where a call's fixed cost goes (configs[2] table shape): propagate time vs packet count, and per wave -- from the pass in
which its first lane found the packet supply empty -- the time to its end, its passes and the live lanes over those passes.
   python tools/exp_drain.py n_packets [n_packets ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine

prob = synthetic.make_problem(seed=1, n_packets=1, n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
BASE = int(os.environ.get("EXP_BASE_FLAGS", "0"))  # e.g. 16777216: fixed cut-offs of the sweep / walk phases (rounds 1-2)
for a in sys.argv[1:]:
    P = int(float(a))
    eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
    out = {}
    for name, flag in (("plain", 0), ("ticks", 2097152), ("passes", 4194304), ("lanes", 8388608), ("all_passes", 32)):
        eng.set_option("debug_flags", flag | BASE)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        out[name] = eng.last_counters()["reserved"]
        if name == "plain":
            kt = eng.last_kernel_times()
    eng.set_option("debug_flags", 0)
    waves = min((P + 63) // 64, 4096)
    c = eng.last_counters()
    print(f"P={P:.0e}: propagate {kt['propagate_ms']:.1f} ms in {kt['launches']} launch(es) = {P / kt['propagate_ms'] / 1e3:.2f} Mpkt/s; per wave after its supply ran out: "
          f"{out['ticks'] / waves / 1e5:.1f} ms, {out['passes'] / waves:.0f} passes of {out['all_passes'] / waves:.0f} "
          f"({1e4 * out['ticks'] / max(out['passes'], 1) / 1e3:.1f} us per drain pass), {out['lanes'] / max(out['passes'], 1):.1f} live lanes per drain pass; "
          f"events/packet {c['events'] / P:.1f}", flush=True)
eng.close()
