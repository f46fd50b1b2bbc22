#!/usr/bin/env python
"""bench.py -- packets/s of the Monte Carlo packet-propagation path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one Monte Carlo iteration of the hot path over one resident batch of synthetic packets: zero the
estimators, propagate every packet of this rank's shard (HIP kernels), and -- for N > 1 -- the one RCCL all-reduce
of the estimator arrays (J, nu_bar, j_blue, Edotlu, v-hist) that an outer plasma iteration needs.  Inputs
(packets, opacity tables) are resident in HBM before the timed region.  Weak scaling: every rank owns
`--packets` packets, so an N-GPU iteration propagates N x packets.

Workload (N = 1): BASELINE.json configs[1] -- tardis_example shape, 1e7 packets, 20 shells, ~3e4 lines,
downbranch, no v-packets, synthetic opacities (the reference's atomic data is not available offline).

The JSON line carries `roofline` (algorithmic bytes of a step's propagation per launch / the HIP-event time of the
dominant kernel -- the propagation kernel; the seeding kernel and the line-estimator passes of the step are reported
beside it -- against the 8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle -- the parity-pinned C port of the reference
algorithm -- timed on this box's cores on a bounded sample of the same workload).
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
from tardis_amd.engine import Engine  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(c: dict, part: str = "step") -> float:
    """48 B per line visit (nu_line 8 + tau 8 + the read-modify-writes of j_blue and Edotlu, 16 each), 56 B
    per event, 8 B per macro-atom transition examined, 16 B per v-packet line visit, 56 B of per-packet I/O.

    part = "step": everything a Monte Carlo iteration moves.  With the wave kernel the line-estimator read-modify-writes
    are not done by the propagation kernel (it logs one record per trace, the accumulate kernel applies them), so the
    dominant kernel's own share is part = "propagate": 16 B per line visit + the rest; part = "estimators": 32 B per visit."""
    per_visit = {"step": 48.0, "propagate": 16.0, "estimators": 32.0}[part]
    if part == "estimators":
        return per_visit * c["line_visits"]
    return (per_visit * c["line_visits"] + 56.0 * c["events"] + 8.0 * c["macro_transitions"] + 16.0 * c["vpacket_line_visits"]
            + 56.0 * c["packets"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config number (2: 1e7 pkts downbranch 3e4 lines)")
    ap.add_argument("--packets", type=int, default=None, help="packets per GPU per step (default: the config's)")
    ap.add_argument("--lines", type=int, default=None)
    ap.add_argument("--shells", type=int, default=None)
    ap.add_argument("--mode", type=str, default=None)
    ap.add_argument("--vpackets", type=int, default=None)
    ap.add_argument("--no-tracking", action="store_true", help="skip the last-interaction tracker outputs")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="packets in the CPU-baseline sample (0: skip)")
    ap.add_argument("--variant", type=int, default=None)
    args = ap.parse_args()

    pg = distributed.init_from_env(backend="gloo")  # control plane only; the data-path collective is RCCL
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
    if args.config in (4, 5) and args.packets is None:
        kw["n_packets"] //= 8  # those configs quote the 8-GPU total
    # every rank draws its own packet shard (iteration = rank changes the packet stream, not the opacities)
    prob = synthetic.make_problem(seed=1, iteration=pg.rank, **kw)
    P = prob.packet_collection.number_of_packets

    eng = Engine(pg.local_rank)
    if args.variant is not None:
        eng.set_option("variant", args.variant)
    eng.set_option("track_last_interaction", 0 if args.no_tracking else 1)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    rccl_ok = distributed.setup_engine_comm(eng, pg)
    if not rccl_ok and pg.rank == 0:
        print("warning: no RCCL communicator; the estimators are all-reduced on the host through gloo", file=sys.stderr)

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
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)  # read after the timed region (event queries would serialise the stream)
    eng.synchronize()
    pg.barrier()
    t1 = time.perf_counter()
    elapsed = pg.max_float(t1 - t0)

    # kernel time of the last step (HIP events on the engine stream) and its work counters
    last_ms = eng.last_propagate_ms()
    ktimes = eng.last_kernel_times()
    res = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    counters = res.counters
    if n_gpus > 1:
        pass  # counters are per rank; every rank runs the same-sized shard

    total_packets = float(P) * n_gpus * args.steps
    value = total_packets / elapsed
    out = {
        "metric": "packets/sec", "value": value, "unit": "packets/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[{args.config - 1}]: tardis_example shape, {P} packets/GPU/step, "
                        f"{kw['n_shells']} shells, {kw['n_lines']} lines, {kw['line_interaction_type']}, "
                        f"{kw.get('n_vpackets', 0)} v-packets, last-interaction tracking "
                        f"{'off' if args.no_tracking else 'on'}; synthetic opacities (SURVEY 8d)",
            "packets_per_gpu": P, "n_shells": kw["n_shells"], "n_lines": kw["n_lines"],
            "line_interaction_type": kw["line_interaction_type"], "n_vpackets": kw.get("n_vpackets", 0),
            "parallelism": (f"packet-sharded x{n_gpus}, " + ("RCCL all-reduce of estimators per step" if rccl_ok
                            else "host (gloo) all-reduce of estimators per step: RCCL communicator unavailable")) if n_gpus > 1 else "single GPU",
        },
    }
    if pg.rank == 0:
        # dominant kernel = the propagation kernel; its launches of one step are timed with HIP events on the engine stream
        launches = max(ktimes["launches"], 1)
        wave = args.variant in (None, 2, 3)
        dominant = "propagate_wave_kernel" if wave else ("propagate_lane_kernel" if args.variant == 0 else "propagate_group_kernel")
        kernel_ms = ktimes["propagate_ms"] / launches
        # the dominant kernel's own algorithmic bytes (see algorithmic_bytes) over its HIP-event duration
        bytes_per_launch = algorithmic_bytes(counters, "propagate" if wave else "step") / launches
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        step_bytes = algorithmic_bytes(counters, "step")
        step_achieved = step_bytes / (last_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        default_workload = (args.config == 2 and args.packets is None and args.lines is None and args.shells is None
                            and args.mode is None and args.vpackets is None and not args.no_tracking and args.variant is None)
        if default_workload and os.path.exists(pmc_path):  # the PMC passes were collected on the default workload
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "kernel": f"{dominant} (dominant kernel of a step)", "kernel_ms": kernel_ms,
                           "launches_per_step": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                           "step": {"algorithmic_bytes": step_bytes, "device_ms": last_ms, "achieved": step_achieved,
                                    "frac": step_achieved / HBM_PEAK_GBS, "seed_kernel_ms": ktimes["seed_ms"],
                                    "estimator_passes_ms": ktimes.get("estimator_ms", 0.0),
                                    "note": "all kernels of one iteration: MT19937 seeding, propagation, line-estimator passes"},
                           "per_packet": {k: counters[k] / max(P, 1) for k in ("line_visits", "events", "macro_transitions", "rng_draws")}}
        if n_gpus == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(prob, eng, min(args.cpu_sample, P))
    if pg.rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    pg.destroy()


def cpu_baseline(prob, eng, n_sample: int) -> dict:
    """Time the CPU oracle (parity-pinned C port of the reference algorithm, OpenMP over packets) on the first
    n_sample packets of the workload, and report the spectrum parity of the GPU result on that sample."""
    from oracle import oracle

    pc = prob.packet_collection
    sub = pc.shard(0, max(pc.number_of_packets // n_sample, 1)) if n_sample < pc.number_of_packets else pc
    n = sub.number_of_packets
    threads = oracle.max_threads()
    args = (sub, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
            prob.spectrum_frequency_grid)
    t0 = time.perf_counter()
    ref = oracle.run(*args, math_mode=oracle.MATH_PORTABLE, n_threads=threads, track_last_interaction=False)
    dt_all = time.perf_counter() - t0
    n1 = max(n // 8, 1)
    sub1 = sub.shard(0, max(n // n1, 1))
    t0 = time.perf_counter()
    oracle.run(sub1, *args[1:], math_mode=oracle.MATH_PORTABLE, n_threads=1, track_last_interaction=False)
    dt_1 = time.perf_counter() - t0
    # GPU result on the same sample (per-packet results do not depend on batching)
    eng.set_packets(sub)
    eng.reset_estimators()
    eng.propagate()
    eng.synchronize()
    got = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    a = spectrum.emitted_luminosity_histogram(got.output_nus, got.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    b = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    return {
        "value": n / dt_all, "unit": "packets/s", "cores": threads, "kind": "port",
        "sample": f"first {n} packets of the workload, CPU oracle (C port of the reference algorithm, -O2 IEEE-strict, "
                  f"OpenMP {threads} threads) {dt_all:.1f} s; 1 thread on {sub1.number_of_packets} packets: "
                  f"{sub1.number_of_packets / dt_1:.0f} packets/s",
        "single_thread_value": sub1.number_of_packets / dt_1,
        "spectrum_rel_l2_gpu_vs_cpu": spectrum.relative_l2(a, b),
        "per_packet_bit_exact": bool(np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)),
        "max_rel_diff_J": float(np.max(np.abs(got.j_estimator - ref.j_estimator) / np.abs(ref.j_estimator))),
        "max_rel_diff_nu_bar": float(np.max(np.abs(got.nu_bar_estimator - ref.nu_bar_estimator) / np.abs(ref.nu_bar_estimator))),
    }


if __name__ == "__main__":
    main()
