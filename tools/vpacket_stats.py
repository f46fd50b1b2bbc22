"""What a v-packet tracer has to schedule, measured on the CPU oracle ("instrument first").

The shipped v-packet path (pooled volleys, csrc/propagate_wave.hpp) keeps ~23 of a wave's 64 lanes busy on the BASELINE
configs[4] table shape.  Question: how much of that is inherent in the work -- v-packets of one volley need very
different numbers of shell crossings -- and how much would a tracer win that takes ONE SHELL CROSSING of ANY v-packet per lane and
round (the dense, asynchronously fed tracer of DESIGN 9-3)?

Method: the oracle logs for every v-packet {shell it starts in, shell crossings traced, lines visited, crossing at which the Russian
roulette dropped it} (trace_vpacket, virtual_packet.py:179-244).  With survival probability 0 a dropped v-packet is dead whatever
its optical depth was, so the engine's prefix-sum screening (csrc/tau_prefix.hpp) makes a crossing O(1): the unit of work is the
crossing.  Models, for the v-packets of N r-packets in launch order:

  volley-synchronous : a wave's lanes each trace one v-packet of the wave's current set of volleys to its end; a round of 64 costs its
                       LONGEST trace (what lock-step lanes pay).  busy = sum of crossings / (64 x sum over rounds of the max)
  pooled, round of R : the shipped scheme -- items of all the wave's volleys are dealt to lanes, R crossings per lane and refill
                       (modelled as rounds of 64 items taken in order, each lane R crossings, leftover crossings re-queued)
  dense              : every lane takes one crossing per round from a common queue: busy = 1 by construction while the queue holds
                       >= 64 items; the cost is a queue operation per crossing

    python tools/vpacket_stats.py [packets=300] > profiles/r04_vpacket_stats.txt
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (analysis tool: the oracle is the trace source here, nothing is measured against it)
from tardis_amd import synthetic  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300
LEVELS = os.environ.get("EXP_LEVELS", "heavy")
kw = dict(synthetic.BASELINE_CONFIGS[5])
kw.pop("n_packets", None)
prob = synthetic.make_problem(seed=1, n_packets=N, level_sizes=LEVELS, **kw)
cap = 1200 * N
buf = np.zeros((cap, 4), dtype=np.int64)
lib = oracle.lib()
lib.oracle_set_vtrace_log.restype = None
lib.oracle_set_vtrace_log.argtypes = [C.c_void_p, C.c_int64]
lib.oracle_vtrace_log_count.restype = C.c_int64
lib.oracle_set_vtrace_log(buf.ctypes.data, cap)
ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                 prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=1, track_last_interaction=False)
n = int(lib.oracle_vtrace_log_count())
lib.oracle_set_vtrace_log(None, 0)
assert n == ref.counters["vpackets"], (n, ref.counters["vpackets"])
v = buf[:n]
shell, cross, visits, dropped = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
S = kw["n_shells"]
print(f"workload: BASELINE configs[4] table shape ({S} shells x {kw['n_lines']} lines, {kw['line_interaction_type']}, {kw['n_vpackets']} v-packets "
      f"per interaction, {LEVELS} levels), {N} packets on the CPU oracle")
print(f"v-packets {n} = {n / N:.0f} per packet in volleys of {kw['n_vpackets']}; lines visited {visits.sum() / n:.0f} per v-packet")
print(f"shell crossings per v-packet: mean {cross.mean():.2f}, median {np.median(cross):.0f}, 90 % {np.percentile(cross, 90):.0f}, "
      f"99 % {np.percentile(cross, 99):.0f}, max {cross.max()}")
esc = dropped < 0
print(f"dropped by the Russian roulette: {(~esc).mean():.3f} of the v-packets, after {cross[~esc].mean():.2f} crossings on average "
      f"(at their first crossing: {(dropped == 1).mean():.3f} of all); escaping: {esc.mean():.3f}, after {cross[esc].mean():.1f} crossings")
print(f"share of the crossings spent on v-packets that end up dropped: {cross[~esc].sum() / cross.sum():.3f}")
hist = np.bincount(np.minimum(cross, 40))
print("crossings histogram (1 .. 39, >= 40):", " ".join(str(x) for x in hist[1:]))
# volleys: consecutive groups of n_v v-packets (the oracle traces a volley's v-packets one after the other)
nv = kw["n_vpackets"]
vol = cross[: (n // nv) * nv].reshape(-1, nv)
print(f"\nwithin a volley ({nv} v-packets of one interaction): max / mean crossings = {vol.max(axis=1).mean() / vol.mean():.2f}; "
      f"volleys whose v-packets all end at their first crossing: {(vol.max(axis=1) == 1).mean():.3f}")


def synchronous(c, width=64):
    m = (len(c) // width) * width
    rounds = c[:m].reshape(-1, width)
    return rounds.sum() / (width * rounds.max(axis=1).sum())


def pooled(c, R, width=64):
    """Lanes take items in order; every refill round each lane advances its item by up to R crossings; finished lanes take the next item
    at the next round.  Returns busy lane-crossings / (width x rounds x R)."""
    queue = list(c)
    lanes = [0] * width
    qi = 0
    rounds = busy = 0
    while qi < len(queue) or any(lanes):
        for k in range(width):
            if lanes[k] == 0 and qi < len(queue):
                lanes[k] = queue[qi]; qi += 1
        for k in range(width):
            d = min(lanes[k], R)
            busy += d
            lanes[k] -= d
        rounds += 1
    return busy / (width * rounds * R)


# a wave's v-packets: those of 64 consecutive r-packets' volleys arrive interleaved; the launch-order log is the serial order -- shuffle
# inside windows of 64 volleys to mimic the interleaving of a wave's lanes
rng = np.random.default_rng(0)
idx = np.arange(len(vol))
for a in range(0, len(idx), 64):
    rng.shuffle(idx[a:a + 64])
stream = vol[idx].reshape(-1)
print("\nlane occupancy of the crossing work (1.0 = every lane advances a live v-packet every round):")
print(f"  volley-synchronous, 64 v-packets per round to their end : {synchronous(stream):.3f}")
for R in (1, 2, 4, 6, 8):
    print(f"  pooled, lanes refilled every {R} crossing(s)            : {pooled(stream[:200_000], R):.3f}")


def phased(c, K, width=64):
    """The wave kernel's volley phase: the K items the wave's owners handed over in one pass are dealt to the lanes (refill after every
    crossing), and the phase -- hence the wave -- goes on until the LAST of them has ended."""
    busy = rounds = 0
    for a in range(0, len(c) - K + 1, K):
        items = c[a:a + K]
        lanes = np.zeros(width, dtype=np.int64)   # greedy list scheduling: the next item goes to the lane that is free first
        for x in items:
            k = int(np.argmin(lanes))
            lanes[k] += x
        busy += int(items.sum()); rounds += int(lanes.max())
    return busy / (width * rounds)


for K in (100, 250, 500, 640):
    print(f"  one phase per {K:3d} items, ends with its last item        : {phased(stream[:150_000], K):.3f}")


def makespan(items, width=64):
    import heapq
    lanes = [0] * width
    for x in items:
        heapq.heappush(lanes, heapq.heappop(lanes) + int(x))
    return max(lanes)


def rounds_of_six(vols, K, asynchronous):
    """The shipped rounds: an owner hands over at most six v-packets at a time (the rest needs the draws the first six consumed).
    Synchronous = the wave's round ends with its longest item, then all owners hand over the rest; asynchronous = an owner hands over its
    rest as soon as ITS six are done (what a per-owner completion counter would buy)."""
    import heapq
    busy = span = 0
    for a in range(0, len(vols) - K + 1, K):
        v = vols[a:a + K]
        busy += int(v.sum())
        if not asynchronous:
            span += makespan(v[:, :6].reshape(-1)) + makespan(v[:, 6:].reshape(-1))
            continue
        lanes = [0] * 64
        done_at = np.zeros(K, dtype=np.int64)
        for o in range(K):          # first rounds, in owner order
            for x in v[o, :6]:
                t = heapq.heappop(lanes) + int(x)
                heapq.heappush(lanes, t)
                done_at[o] = max(done_at[o], t)
        for o in np.argsort(done_at):  # second rounds, released when the owner's first round is complete
            for x in v[o, 6:]:
                t = max(heapq.heappop(lanes), int(done_at[o])) + int(x)
                heapq.heappush(lanes, t)
        span += max(lanes)
    return busy / (64 * span)




def carry_over(vols, K, cut, width=64):
    """Rounds of <= 6 per owner as shipped, but a volley phase ends as soon as no item is waiting and at most `cut` lanes still trace:
    those items stay with their lanes into the NEXT pass's phase, and their owners wait for them (no event, no new volley) while the other
    owners go on.  K owner slots; every pass each free slot starts the next volley of the stream.  Time-stepped: one crossing per busy
    lane and step; only the volley phases are counted."""
    import collections
    busy = steps = 0
    lanes = np.zeros(width, dtype=np.int64)          # crossings left of the item a lane holds
    lane_owner = -np.ones(width, dtype=np.int64)
    left = [[] for _ in range(K)]                    # per owner slot: items of its volley not yet handed over
    out = np.zeros(K, dtype=np.int64)                # per owner slot: items of its current round still tracing
    nxt = 0
    while nxt < len(vols) or any(left) or lanes.any():
        for o in range(K):                           # the event phase: owners without an unfinished volley interact again
            if out[o] == 0 and not left[o] and nxt < len(vols):
                left[o] = list(vols[nxt]); nxt += 1
        while True:                                  # the volley phase of this pass: rounds until the cut-off
            queue = collections.deque()
            for o in range(K):
                if out[o] == 0 and left[o]:
                    take, left[o] = left[o][:6], left[o][6:]
                    out[o] = len(take)
                    queue.extend((o, x) for x in take)
            if not queue and not lanes.any():
                break
            while True:
                for k in np.flatnonzero(lanes == 0):
                    if not queue:
                        break
                    o, x = queue.popleft()
                    lanes[k] = x; lane_owner[k] = o
                n_busy = int((lanes > 0).sum())
                more = nxt < len(vols) and any(out[o] == 0 and not left[o] for o in range(K))   # an owner that can go on exists
                if n_busy == 0 or (not queue and n_busy <= cut and more):
                    break
                busy += n_busy; steps += 1
                lanes[lanes > 0] -= 1
                for k in np.flatnonzero((lanes == 0) & (lane_owner >= 0)):
                    out[lane_owner[k]] -= 1; lane_owner[k] = -1
            if not any(out[o] == 0 and left[o] for o in range(K)):
                break                                # nobody can hand over another round now: on to the next pass
    return busy / (width * steps)


volw = vol[idx]
for K in (20, 43):
    print(f"  rounds of <= 6 + carry-over, {K:2d} owner slots: phase ends at 0 / <= 8 / <= 16 tracing lanes "
          f"{carry_over(volw[:6_000], K, 0):.3f} / {carry_over(volw[:6_000], K, 8):.3f} / {carry_over(volw[:6_000], K, 16):.3f}")
for K in (20, 43, 64):
    print(f"  rounds of <= 6 per owner, {K:2d} owners per pass: wave-synchronous {rounds_of_six(volw[:12_000], K, False):.3f}, "
          f"per-owner asynchronous {rounds_of_six(volw[:12_000], K, True):.3f}")
print("  dense (one crossing per lane and round from one queue)  : 1.000 by construction; 1 queue operation per crossing")
print(f"""
Reading.  (1) The work is short and skewed: three crossings per v-packet on average, a third of the v-packets end at their first
crossing, one in a hundred needs 15 or more -- lanes that trace "their" v-packet to its end in lock-step are busy {synchronous(stream):.2f} of the time
(the group kernel's volley phase: ~12 of 64 lanes live, DESIGN 5.2c).  (2) The skew itself is NOT what limits a tracer that refills a
lane after every crossing: {pooled(stream[:200_000], 1):.2f}.  The pooled volleys of the wave kernel are built that way and still report 23 of 64 lanes
(0.36): their loss is the PHASE -- a wave's 64 owners hand over ~{int(round(64 * len(vol) / max(ref.counters["events"], 1))) * nv} items per pass (one volley per interaction), and the volley phase
lasts until the longest of them has ended (the "one phase per K items" rows: a 30-crossing item pins the wave while 63 lanes idle) --
and speculation: an item's position in its owner's random stream is predicted (1.32 traces per committed v-packet, at most six items
per owner and round -- letting every owner start its second round as soon as ITS first is complete would add only ~0.05, last rows).
What would pay inside the wave kernel is a CUT-OFF with carry-over, as the sweeps and the macro-atom walks already have: end the volley
phase once nothing waits and <= 16 lanes still trace, let those lanes keep their items into the next pass's phase and their owners wait
(the "carry-over" rows: x1.2 - 1.3 at <= 8 lanes, x1.4 - 1.5 at <= 16).
(3) A dense tracer fed from ALL waves (DESIGN 9-3) removes both; its price is one queue operation per crossing and
the hand-back of the draws a volley consumed.  Upper bound of the v-packet part's speed-up from occupancy
alone: x{1 / 0.36:.1f} on ~95 % of a configs[4]-shape step.""")
