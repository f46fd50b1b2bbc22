"""Cross-packet locality of the line sweeps, measured on the CPU oracle's traces ("instrument first").

Question: if the lanes of a wave (a CU, an XCD) were (re)filled from queues binned by (shell, line position), how many of the
64-byte sectors of the tau table / the line list their sweeps read would they SHARE?  That bounds what any re-binning scheme
can save, before its own cost (moving a packet between lanes through memory at every event).

Method: the oracle logs every trace {shell, first line, lines visited} of N packets on the BASELINE configs[2] table shape.
A snapshot of the chip "at one moment" is a uniform sample of as many traces as the kernel has lanes in flight (a lane is
always inside some trace of some packet; traces are sampled in proportion to their number of 8-line sweep rounds, i.e. to
the time a lane spends in them).  For groups of 64 (wave), 1024 (CU: 16 waves), 32 768 (XCD: one L2) lanes -- as launched
(arbitrary neighbours) and sorted by (shell, first line) -- count the distinct tau sectors the group's traces touch against
the sum over its lanes.  sharing = 1 - distinct / sum.

    python tools/locality_stats.py [packets=4000] [lanes=262144]  > profiles/r04_locality_stats.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (analysis tool: the oracle is the trace source here, nothing is measured against it)
from tardis_amd import synthetic  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4000
LANES = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4096 * 64
LEVELS = os.environ.get("EXP_LEVELS", "uniform")
S, L = 20, 500_000
prob = synthetic.make_problem(seed=1, n_packets=N, n_shells=S, n_lines=L, line_interaction_type="macroatom", level_sizes=LEVELS)
cap = 400 * N
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
tr = buf[:n]
shell, start, cnt = tr[:, 0], tr[:, 1], tr[:, 2]
print(f"workload: BASELINE configs[2] table shape ({S} shells x {L} lines, macroatom, {LEVELS} levels), {N} packets on the CPU oracle")
print(f"traces {n} = {n / N:.1f} per packet; lines visited per trace: mean {cnt.mean():.1f}, median {np.median(cnt):.0f}, "
      f"90 % {np.percentile(cnt, 90):.0f}, 99 % {np.percentile(cnt, 99):.0f}, max {cnt.max()}")
sh_share = np.bincount(shell, minlength=S) / n
print("share of the traces per shell:", " ".join(f"{v:.3f}" for v in sh_share))
# sectors a sweep of the lane kernel reads: 8-line chunks from the trace's first line to its last, 64-byte sectors of the row
first_sec = (shell * L + start) >> 3
last_sec = (shell * L + start + np.maximum(cnt, 1) - 1 + 7) >> 3  # (+7: the last chunk is read whole)
nsec = last_sec - first_sec + 1
rounds = (np.maximum(cnt, 1) + 7) // 8
print(f"tau sectors per trace (8-line chunks, 64-byte sectors): mean {nsec.mean():.2f}; sweep rounds per trace: mean {rounds.mean():.2f}")
rng = np.random.default_rng(0)
p = rounds / rounds.sum()
print(f"\nsnapshot: {LANES} lanes in flight, each inside a trace drawn in proportion to its sweep rounds")
print(f"{'group':>22s} {'lanes':>7s} {'sum of sectors':>15s} {'distinct':>10s} {'sharing':>8s}   {'window sharing*':>15s}")
for rep in range(1):
    pick = rng.choice(n, size=LANES, p=p)
    f, l_ = first_sec[pick], last_sec[pick]
    order_sorted = np.argsort(f, kind="stable")
    for label, order in (("as launched", np.arange(LANES)), ("sorted (shell, line)", order_sorted)):
        for g in (64, 1024, 32768, LANES):
            tot = dist = 0
            tot_w = dist_w = 0
            for a in range(0, LANES, g):
                idx = order[a:a + g]
                # distinct sectors of the union of the intervals [f, l]
                ff, ll = f[idx], l_[idx]
                o = np.argsort(ff, kind="stable")
                ff, ll = ff[o], ll[o]
                run_end = np.maximum.accumulate(ll)
                new_start = np.ones(len(ff), bool)
                new_start[1:] = ff[1:] > run_end[:-1]
                # union length = sum over merged runs
                starts = ff[new_start]
                ends = np.maximum.reduceat(ll, np.flatnonzero(new_start))
                dist += int((ends - starts + 1).sum())
                tot += int((ll - ff + 1).sum())
                # *window sharing: what is shared AT ONE MOMENT -- every lane is at one chunk (one sector) of its trace; distinct
                # current sectors of the group, lanes at a uniformly random round of their trace
                cur = ff + (rng.random(len(ff)) * (ll - ff + 1)).astype(np.int64)
                dist_w += len(np.unique(cur))
                tot_w += len(cur)
            print(f"{label:>22s} {g:7d} {tot:15d} {dist:10d} {1 - dist / tot:8.3f}   {1 - dist_w / tot_w:15.3f}")
print("""
sharing         = share of the group's tau-sector reads that another lane of the group also makes during its CURRENT trace
                  (an upper bound of what a group-wide cache of unlimited size, filled for the duration of one trace, could save);
window sharing* = the same for the ONE sector every lane reads in the current sweep round (what lanes of a wave issuing their
                  chunk loads together could coalesce).
Cells: 20 shells x 62 500 sectors per row = 1.25e6 tau sectors; 262 144 lanes x ~5.5 sectors per trace = 1.4e6 sector reads per
"generation" of traces -- about one read per sector of the table: even perfectly sorted neighbours mostly read different sectors.""")
