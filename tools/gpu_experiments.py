"""GPU-box experiment driver (design input, not a test): memory-system micro-benchmarks and kernel-time ablations.
    python tools/gpu_experiments.py [micro] [ablate] [scale]
Writes human-readable tables to stdout."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

what = set(sys.argv[1:]) or {"micro", "ablate"}
eng = Engine(0)

if "micro" in what:
    names = {0: "atomic f64 random, agent scope", 1: "atomic f64 random, wg scope, XCD slice",
             2: "atomic f64 16-coalesced, agent", 3: "atomic f64 16-coalesced, wg scope, XCD slice",
             4: "load 8B random", 5: "load 8B 16-coalesced"}
    blocks, iters = 2048, 256
    nops = blocks * 256 * iters
    for n in (600_000, 20_000_000, 200_000_000):
        for which in range(6):
            ms = eng.debug_microbench(which, n, iters, blocks)
            print(f"micro n={n * 8 / 1e6:8.1f} MB  {names[which]:46s} {ms:9.3f} ms  {nops / ms / 1e6:9.2f} Gop/s", flush=True)

def run(P, flags=0, bpc=16, copies=1, track=1, kw=None, reps=2, variant=1, G=16, occ=4):
    kw = kw or dict(n_shells=20, n_lines=30000, line_interaction_type="downbranch")
    prob = run.cache.get((P, tuple(sorted(kw.items()))))
    if prob is None:
        prob = synthetic.make_problem(seed=1, n_packets=P, **kw)
        run.cache[(P, tuple(sorted(kw.items())))] = prob
    eng.set_option("variant", variant)
    eng.set_option("group_size", G)
    eng.set_option("waves_per_simd", occ)
    eng.set_option("debug_flags", flags)
    eng.set_option("blocks_per_cu", bpc)
    eng.set_option("estimator_copies", copies)
    eng.set_option("track_last_interaction", track)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    best = 1e30
    for _ in range(reps):
        eng.reset_estimators()
        eng.propagate()
        eng.synchronize()
        best = min(best, eng.last_propagate_ms())
    return best
run.cache = {}

if "ablate" in what:
    for P in (1_000_000, 10_000_000):
        ms = run(P, variant=0, bpc=8)
        print(f"ablate P={P:>9d} variant=0: {ms:9.2f} ms  {P / ms / 1e3:8.2f} Mpkt/s", flush=True)
        for G, occ in ((16, 4), (8, 4)):
            ms = run(P, G=G, occ=occ)
            print(f"ablate P={P:>9d} variant=1 G={G:2d} occ={occ}: {ms:9.2f} ms  {P / ms / 1e3:8.2f} Mpkt/s", flush=True)
        for flags in (1, 4):
            ms = run(P, flags, G=8)
            print(f"ablate P={P:>9d} variant=1 G=8 occ=4 flags={flags}: {ms:9.2f} ms  {P / ms / 1e3:8.2f} Mpkt/s", flush=True)
        ms = run(P, track=0)
        print(f"ablate P={P:>9d} variant=1 G=16 occ=3 track=0: {ms:9.2f} ms  {P / ms / 1e3:8.2f} Mpkt/s", flush=True)
        for bpc in (1, 2, 4, 6):
            ms = run(P, bpc=bpc)
            print(f"ablate P={P:>9d} variant=1 G=16 occ=3 blocks/CU={bpc}: {ms:9.2f} ms  {P / ms / 1e3:8.2f} Mpkt/s", flush=True)

if "scale" in what:
    kw = dict(n_shells=20, n_lines=500000, line_interaction_type="macroatom")
    for P in (200_000, 2_000_000):
        for G, occ in ((16, 4), (8, 4)):
            ms = run(P, kw=kw, reps=1, G=G, occ=occ)
            print(f"config3-shape P={P} variant=1 G={G} occ={occ}: {ms:9.2f} ms {P / ms / 1e3:8.3f} Mpkt/s", flush=True)
pass

if "vpk" in what:
    eng2 = Engine(0)
    for kw, P in ((dict(n_shells=20, n_lines=30000, line_interaction_type="downbranch", n_vpackets=10), 2_000_000),
                  (dict(n_shells=20, n_lines=30000, line_interaction_type="macroatom", n_vpackets=3), 200_000)):
        prob = synthetic.make_problem(seed=1, n_packets=P, **kw)
        eng2.set_geometry(prob.geometry, prob.time_explosion)
        eng2.set_opacity(prob.opacity_state)
        eng2.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        eng2.set_packets(prob.packet_collection)
        for _ in range(2):
            eng2.reset_estimators(); eng2.propagate(); eng2.synchronize()
        ms = eng2.last_propagate_ms()
        res = eng2.get_results(track_last_interaction=False, want_line_estimators=False)
        c = res.counters
        print(f"vpackets {kw['line_interaction_type']} n_v={kw['n_vpackets']} P={P}: {ms:9.2f} ms {P / ms / 1e3:8.3f} Mpkt/s; per packet: "
              f"vpackets={c['vpackets'] / P:.1f} vp_line_visits={c['vpacket_line_visits'] / P:.0f} line_visits={c['line_visits'] / P:.0f}; "
              f"{c['vpacket_line_visits'] / ms / 1e6:.2f} G vp-line-visits/s; traced/committed={c['reserved'] / max(c['vpacket_line_visits'], 1):.2f}", flush=True)
    eng2.close()
eng.close()
