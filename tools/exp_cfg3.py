"""Timing experiments on the BASELINE configs[2] table shape (GPU box): one engine, a list of option sets.
    python tools/exp_cfg3.py [packets] [name=value,name=value ...]...
Each argument after the packet count is one experiment: comma-separated engine options (variant, debug_flags, group_size,
ls_min_active, ls_max_steps, waves_per_simd, blocks_per_cu, track_last_interaction ...)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
exps = sys.argv[2:] or ["variant=-1"]
shape = dict(n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
if os.environ.get("EXP_SHAPE") == "config2":
    shape = dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
if os.environ.get("EXP_SHAPE") == "config2v":
    shape = dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=10)
if os.environ.get("EXP_SHAPE") == "config5":
    shape = dict(n_shells=100, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=10)
if os.environ.get("EXP_NV") is not None:
    shape["n_vpackets"] = int(os.environ["EXP_NV"])
if os.environ.get("EXP_LOG_TAU"):  # thinner / thicker lines than the default synthetic ejecta (log10 of the mean Sobolev depth)
    shape["log_tau_mean"] = float(os.environ["EXP_LOG_TAU"])
if os.environ.get("EXP_NE0"):
    shape["electron_density_0"] = float(os.environ["EXP_NE0"])
if os.environ.get("EXP_LEVELS"):  # "heavy": heavy-tailed macro-atom blocks (synthetic.make_opacity_state)
    shape["level_sizes"] = os.environ["EXP_LEVELS"]
prob = synthetic.make_problem(seed=1, n_packets=1, **shape)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
defaults = {"log_tail_split": 1, "log_tail_packets": 8, "variant": -1, "debug_flags": 0, "group_size": 0, "lane_sweep_min_active": -1, "walk_min_active": -2, "lane_sweep_max_steps": 1 << 30, "waves_per_simd": 4,
            "track_last_interaction": 1, "walk_hot": -1, "walk_hot_min_mass": int(os.environ.get("EXP_HOT_SHORT", 800)),
            "walk_hot_min_mass_long": int(os.environ.get("EXP_HOT_LONG", 400)), "walk_sector_packing": 1,
            "vp_carry_min_active": 16, "pass_cus": 0, "bucket_lines_permille": 750, "vpk_wide_registers": 1, "ls_waves_per_simd": 4, "sweep_table": -1, "log_sets": 0, "est_accumulate": 3, "est_pipeline": 1, "epoch_split": 0, "log_by_shell": 0, "drain_compact": 0, "drain_split": 0, "drain_pack_lanes": 64, "est_one_level": 1}  # (4: the 128-VGPR lane-sweep instantiation, so that A/B lines compare like with like; 0 = the engine's own choice)
TABLE_OPTIONS = ("walk_hot", "walk_hot_min_mass", "walk_hot_min_mass_long", "walk_sector_packing", "bucket_lines_permille")  # take effect in set_opacity
table_state = (-1, 800, 400, 1, 750)  # the engine's defaults, in force for the set_opacity above
ref = None
for e in exps:
    opts = dict(defaults)
    for kv in e.split(","):
        if kv:
            k, v = kv.split("=")
            opts[k] = int(v)
    for k, v in opts.items():
        try:
            eng.set_option(k, v)
        except RuntimeError:  # (an older build of the library, loaded through TARDIS_MC_LIB for an A/B, does not know the newer options)
            if k in e:
                raise
    ts = tuple(opts[k] for k in TABLE_OPTIONS)
    if ts != table_state and table_state is not None:
        eng.set_opacity(prob.opacity_state)
    table_state = ts
    eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
    best = 1e30
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        best = min(best, eng.last_propagate_ms())
    kt = eng.last_kernel_times()
    c = eng.last_counters()
    est = ""
    if os.environ.get("EXP_EST_SUMS"):  # the line estimators themselves: sums over all (line, shell) cells and an index-weighted sum
        import numpy as np
        r = eng.get_results(track_last_interaction=False, want_packet_outputs=False)
        wgt = np.arange(r.j_blue_estimator.size, dtype=np.float64).reshape(r.j_blue_estimator.shape) % 977.0
        est = f"  jb {r.j_blue_estimator.sum():.12e} {(r.j_blue_estimator * wgt).sum():.12e} ed {r.edotlu_estimator.sum():.12e} {(r.edotlu_estimator * wgt).sum():.12e}"
    sig = (c["line_visits"], c["events"], c["macro_transitions"], c["rng_draws"])
    if ref is None:
        ref = sig
    vp = f" vp/pkt {c['vpackets'] / P:7.1f} vpvis/vp {c['vpacket_line_visits'] / max(c['vpackets'], 1):7.1f}" if c.get("vpackets") else ""
    print(f"{e:60s} {best:9.2f} ms  {P / best / 1e3:7.2f} Mpkt/s  propagate {kt['propagate_ms']:8.2f} ms x{kt['launches']}  "
          f"est {kt['estimator_ms']:7.2f} ms  packed x{eng.last_compactions() if hasattr(eng, 'last_compactions') else 0}  counters {'same' if sig == ref else 'DIFFER ' + str(sig)}  c7={c['reserved']}{vp}{est}", flush=True)
eng.close()
