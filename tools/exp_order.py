"""Does the ORDER in which a call's packets are handed out change its drain?  (GPU box.)  Packets are handed out by index; a packet's results do not depend on the
order.  The same packets are given to the engine as drawn, and sorted by a guess of their lifetime (initial nu, initial mu) so that the likely long-lived ones start first.
    python tools/exp_order.py [packets] [order ...]      orders: asis nu_desc nu_asc mu_asc mu_desc oracle_desc oracle_asc"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import state as st, synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000
orders = sys.argv[2:] or ["asis", "nu_desc", "nu_asc", "mu_asc", "mu_desc", "oracle_desc", "oracle_asc", "asis"]
kw = dict(synthetic.BASELINE_CONFIGS[3]); kw.pop("n_packets")
prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes="heavy", **kw)
pc0 = synthetic.black_body_packets(n, float(prob.geometry.r_inner[0]), 1.0e4)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.set_option("ls_waves_per_simd", 4)
counts = None
for order in orders:
    if order == "asis":
        perm = np.arange(n)
    elif order.startswith("nu_"):
        perm = np.argsort(pc0.initial_nus, kind="stable")
    elif order.startswith("mu_"):
        perm = np.argsort(pc0.initial_mus, kind="stable")
    else:  # the packets' own event counts of an earlier run: what a perfect guess would give
        perm = np.argsort(counts, kind="stable")
    if order.endswith("_desc"):
        perm = perm[::-1]
    pc = st.PacketCollection(pc0.initial_radii[perm], pc0.initial_nus[perm], pc0.initial_mus[perm], pc0.initial_energies[perm], pc0.packet_seeds[perm],
                             pc0.radiation_field_luminosity)
    eng.set_packets(pc)
    best = 1e30
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        best = min(best, eng.last_propagate_ms())
    kt = eng.last_kernel_times()
    res = eng.get_results(track_last_interaction=True, want_line_estimators=False)
    ic = res.trackers.interactions_count
    if counts is None:
        counts = np.empty(n, dtype=np.int64); counts[perm] = ic
        r = np.corrcoef(np.log(pc0.initial_nus), np.log1p(counts))[0, 1]
        top = np.argsort(counts)[-1000:]
        print(f"   events: mean {counts.mean():.1f} max {counts.max()}  corr(log nu0, log events) {r:.3f}   nu0 rank of the 1000 longest (0 = lowest nu): "
              f"median {np.median(np.argsort(np.argsort(pc0.initial_nus))[top]) / n:.3f}   mu0 median {np.median(pc0.initial_mus[top]):.3f}")
    print(f"{order:12s} {best:9.2f} ms   propagate {kt['propagate_ms']:9.2f} ms x{kt['launches']}   passes {kt['estimator_ms']:8.2f}   max events {ic.max()}", flush=True)
eng.close()
