"""What the lane sweep of the headline kernel costs in L1 (vector memory) accesses, and what a sweep on packed conservative bounds would
change -- a CPU study on the oracle's traces (round 5; read together with the TA / TCP counters of profiles/r05_config3_rocprof.txt).

The kernel's lane sweep reads, per lane and step, eight line frequencies and eight optical depths as fp64: 2 x 64 bytes = eight 16-byte
loads, each lane at its own address -> eight L1 accesses per lane and step.  A sweep only has to PROVE that a line does not stop the trace
(propagate_wave.hpp, "lane sweep"); the stopping line is evaluated with the reference's arithmetic.  Bounds survive rounding the table
entries outwards, so the proof could run on a packed table (nu rounded down, tau rounded up, 4 + 4 bytes per line and shell): sixteen lines
per step for the same eight loads.  What that cannot give is the exact serial optical depth in front of the stopping line, which the
reference's outcome needs (i) to decide a stop that is within the bounds' slack and (ii) as an OUTPUT when the trace ends in electron
scattering (d_continuum = (tau_event - tau_prev) / chi).  This script counts how often each case occurs.

    EXP_LEVELS=heavy python tools/sweep_study.py [packets=20000]  > profiles/r05_sweep_study.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (analysis tool: the oracle is the trace source here, nothing is measured against it)
from tardis_amd import synthetic  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
LEVELS = os.environ.get("EXP_LEVELS", "heavy")
S, L = 20, 500_000
prob = synthetic.make_problem(seed=1, n_packets=N, n_shells=S, n_lines=L, line_interaction_type="macroatom", level_sizes=LEVELS)
cap = 600 * N
buf = np.zeros((cap, 4), dtype=np.int64)
lib = oracle.lib()
lib.oracle_set_trace_log.restype = None
lib.oracle_set_trace_log.argtypes = [oracle.C.c_void_p, oracle.C.c_int64]
lib.oracle_trace_log_count.restype = oracle.C.c_int64
lib.oracle_set_trace_log(buf.ctypes.data, cap)
ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                 prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=1, track_last_interaction=False)
n = int(lib.oracle_trace_log_count())
lib.oracle_set_trace_log(None, 0)
assert n < cap
tr = buf[:n]
cnt, typ = np.maximum(tr[:, 2], 1), tr[:, 3]
c = ref.counters
print(f"workload: BASELINE configs[2] tables ({S} shells x {L} lines, macroatom, {LEVELS} levels), {N} packets on the CPU oracle (serial run)")
print(f"traces {n} = {n / N:.1f} per packet; line visits {c['line_visits'] / N:.0f} per packet; lines per trace: mean {cnt.mean():.1f}, median {np.median(cnt):.0f}, "
      f"90 % {np.percentile(cnt, 90):.0f}, 99 % {np.percentile(cnt, 99):.0f}, max {cnt.max()}")
names = {1: "boundary", 2: "line", 4: "electron scattering"}  # InteractionType (oracle: IT_*)
vals, counts = np.unique(typ, return_counts=True)
print("traces by what ends them: " + ", ".join(f"{names.get(int(v), str(int(v)))} {k / n:.3f} (lines per trace {cnt[typ == v].mean():.1f})" for v, k in zip(vals, counts)))
print()
print("steps and 16-byte loads per lane (each one L1 access) of a trace, by lines per step and bytes per line:")
print(f"{'lines/step':>10s} {'bytes/line':>10s} {'steps per trace':>16s} {'loads per trace':>16s} {'loads per packet':>17s}")
base = None
for lines, bpl in ((8, 16), (12, 16), (16, 16), (16, 8), (32, 8)):
    steps = (cnt + lines - 1) // lines
    loads = steps * (lines * bpl // 16)
    if base is None:
        base = loads.sum()
    print(f"{lines:10d} {bpl:10d} {steps.mean():16.2f} {loads.mean():16.2f} {loads.sum() / N:17.0f}   ({loads.sum() / base:.2f} x)")
print()
esc = typ == 4
print("a packed sweep's second pass (exact serial sums from the fp64 table up to the stopping line) is needed for the traces that end in electron scattering")
print(f"  -- {esc.mean():.3f} of the traces, {cnt[esc].sum() / cnt.sum():.3f} of the line visits -- and for optical-depth decisions inside the bounds' slack (2^-23 relative of the running")
print("  sum: the threshold -log(xi) is continuous, ~1e-7 per line visit, ~4e-6 of the traces).  The frequency slack (2^-24 nu_line against ~9e-6 nu_line between neighbouring lines)")
print("  only makes ~0.7 % of the boundary stops evaluate one more line exactly.")
for lines in (16, 32):
    steps1 = (cnt + lines - 1) // lines
    loads = steps1 * (lines * 8 // 16) + np.where(esc, ((cnt + 7) // 8) * 8, 0)
    print(f"  packed {lines} lines per step + fp64 second pass of the electron-scattering traces: {loads.sum() / N:.0f} loads per packet ({loads.sum() / base:.2f} x), "
          f"{(steps1 + np.where(esc, (cnt + 7) // 8, 0)).mean():.2f} dependent steps per trace (now {((cnt + 7) // 8).mean():.2f})")
