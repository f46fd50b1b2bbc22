"""Randomised parity campaign (GPU box): random table shapes, line modes and engine options against the CPU oracle -- per-packet results bit for bit, counters
exactly, estimators to the summation-order tolerance.  A tool, not a test: the fixed cases live in tests/.
    python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (tools/ may use the checker)
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
oracle.build()
bad = 0
t_start = time.perf_counter()
for case in range(n_cases):
    n_lines = int(rng.choice([5, 9, 17, 63, 500, 2_000, 30_000, 200_000]))
    n_lines += int(rng.integers(0, 8)) if n_lines > 8 else 0
    kw = dict(seed=int(rng.integers(1, 1 << 30)), n_packets=int(rng.integers(3_000, 40_000)), n_shells=int(rng.integers(1, 41)), n_lines=n_lines,
              line_interaction_type=str(rng.choice(["scatter", "downbranch", "macroatom"])), level_sizes=str(rng.choice(["uniform", "heavy"])),
              log_tau_mean=float(rng.uniform(-6.0, -2.0)), electron_density_0=float(10.0 ** rng.uniform(7.5, 10.0)))
    if n_lines < 64:
        kw["level_sizes"] = "uniform"
    opts = dict(sweep_table=int(rng.choice([-1, -1, 2, 0])), ls_waves_per_simd=int(rng.choice([0, 3, 4])), est_accumulate=int(rng.choice([3, 3, 2, 1])),
                log_by_shell=int(rng.choice([0, 0, 1])), log_sets=int(rng.choice([0, 1, 2])))
    if rng.random() < 0.4:
        opts["log_capacity"] = int(rng.integers(1 << 16, 1 << 20))
    if rng.random() < 0.3:  # (the packed drain: needs two sets and the chunk-pool log)
        opts.update(drain_compact=int(rng.choice([2, 8, 16, 40])), drain_pack_lanes=int(rng.choice([64, 64, 16, 3])), log_by_shell=0, log_sets=2)
    stream = rng.random() < 0.4  # result streaming: the caller's arrays registered before the call
    if stream:
        opts["stream_min_packets"] = int(rng.choice([64, 1024, 8192]))
    try:
        prob = synthetic.make_problem(**kw)
    except Exception as exc:  # noqa: BLE001 -- a shape the generator does not make
        print(f"case {case}: generator: {exc}")
        continue
    ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                     prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())
    with Engine(0) as eng:
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators()
        if stream:
            from tardis_amd import state as st
            P = kw["n_packets"]
            o_nu, o_en, trk = np.full(P, -7.0), np.full(P, -7.0), st.LastInteractionTrackers(P)
            eng.stream_results(o_nu, o_en, trk)
            eng.propagate(); eng.synchronize()
            got = eng.get_results(o_nu, o_en, track_last_interaction=True, trackers=trk)
            opts["streamed"] = eng.streamed_packets()
        else:
            eng.propagate(); eng.synchronize()
            got = eng.get_results(track_last_interaction=True)
        opts["launches"] = eng.last_kernel_times()["launches"]; opts["packed"] = eng.last_compactions()
        variant = eng.last_variant()
    ok = np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    ok = ok and all(got.counters[k] == ref.counters[k] for k in ("line_visits", "events", "macro_transitions", "rng_draws"))
    ok = ok and np.array_equal(got.trackers.interactions_count, ref.trackers.interactions_count)
    ok = ok and np.array_equal(got.trackers.interaction_line_emit_id, ref.trackers.interaction_line_emit_id)
    ok = ok and np.array_equal(got.trackers.after_nu, ref.trackers.after_nu, equal_nan=True) and np.array_equal(got.trackers.shell_id, ref.trackers.shell_id)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.nanmax(np.where(ref.j_blue_estimator != 0, np.abs(got.j_blue_estimator - ref.j_blue_estimator) / np.abs(ref.j_blue_estimator), 0.0))
        zero_ok = np.array_equal(got.j_blue_estimator == 0, ref.j_blue_estimator == 0)
        relj = np.max(np.abs(got.j_estimator - ref.j_estimator) / np.abs(ref.j_estimator))
    ok = ok and rel < 1e-11 and relj < 1e-11 and zero_ok
    bad += 0 if ok else 1
    print(f"case {case:3d} {'ok ' if ok else 'BAD'} variant {variant} L={kw['n_lines']} S={kw['n_shells']} {kw['line_interaction_type']:10s} {kw['level_sizes']:7s} P={kw['n_packets']} "
          f"events/pkt {ref.counters['events'] / kw['n_packets']:.1f} opts {opts}  jblue rel {rel:.1e}", flush=True)
print(f"{n_cases} cases, {bad} bad, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if bad else 0)
