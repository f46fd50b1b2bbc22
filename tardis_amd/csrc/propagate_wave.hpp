// propagate_wave.hpp -- wave-owner propagation kernel (variants 2 and 3): every LANE owns a packet.
//
// Why (measured on the group kernel, profiles/r01_*): with one packet per G-lane group, (a) the packet's scalar event
// code (boundary distance, tau_event, move, scatter, macro atom) is executed redundantly by all G lanes, i.e. a
// 64-lane wave retires 64/G events per pass through ~700 instructions, and (b) the wave's groups sweep in lockstep, so
// every trace costs the MAXIMUM number of G-line steps over the wave's groups.  Here the two kinds of work are separated:
//
//   event phase  -- lane-per-packet: all 64 lanes run the scalar event code for 64 different packets at once (log
//                   record and epilogue of the finished trace, macro-atom jump, packet hand-over, prologue of the next trace).
//   sweep phase  -- lane sweeps (LS instantiations, partial relativity): every lane sweeps the line list of its own packet,
//                   eight lines per step, proving for each line that the reference's stop tests come out negative and
//                   evaluating only the stopping line with the reference's arithmetic (see lane_exact_line below);
//                -- group sweeps (full relativity): the 64/G groups take the prepared traces from the wave's pool one
//                   after the other, a group that finds its stopping line continues with the next waiting packet.
//
// The arithmetic that decides a trace is the reference's (same operation order, same serial optical-depth sum), so
// per-packet results stay bit-identical to the CPU oracle.  The line estimators are not updated by the sweep at all: the
// owner lane logs one record per trace and the kernels of estimator_log.hpp turn the log into j_blue / Edotlu (the
// memory-side fp64 atomics were what bounded the group kernel).  MT19937: launch_prep_kernel precomputes one word of the
// start state per packet, a lane regenerates the next 8 words of its packet's state itself (refill) and parks the tempered
// doubles in a per-lane LDS ring from which the event code pops its draws.  The kernel is bound by dependent memory
// round trips: most of its structure exists to have fewer of them per pass.
#pragma once
#include "mc_device.hpp"
#include "propagate_group.hpp"
#include "estimator_log.hpp"
#include "walk_tables.hpp"

namespace mc {

// (Cache-policy hints on the kernel's streaming data -- non-temporal / write-through stores of log, tracker and MT19937 words,
// non-temporal loads of the tau chunks, the walk tables and the MT19937 state -- were measured in round 3 behind run-time flags and
// removed again: all neutral or slower, profiles/r03_cache_policy_ab.txt)
constexpr int WV_STATE_STRIDE = 624;  // words between the MT19937 states of two packets
constexpr int WV_RING = 8, WV_RING_VPK = 16;  // look-ahead doubles per packet (power of two; a refill adds 4, so it needs r_cnt <= 4).  16 with 8-double
                            // refills was measured: fewer refill rounds, but the extra 4 KiB of LDS costs the 12th wave of the CU
enum : int { WS_NEED_PACKET = 0, WS_NEED_TRACE = 1, WS_SWEEP = 2, WS_DONE = 3, WS_WALK = 4, WS_VOLLEY = 5, WS_VCARRY = 6 };  // WS_WALK: a macro-atom walk carried over to the next pass;
                                                                                                              // WS_VOLLEY: a round of the packet's volley is with the v-packet tracer (volley queue); WS_VCARRY: a round of its pooled volley was
                                                                                                              // carried over to the next pass (VpPark)
constexpr int RES_PENDING = -1;
// debug_flags bits that read the kernel's profiling / test counters or ablation switches: those are only compiled into the cross-check instantiations (XWALK, the
// `DBG` constant of propagate_wave_kernel), so the host launches one of those whenever any of these bits is set -- ONE list for both sides
constexpr int WV_DBG_FLAGS = 1 | 2 | 4 | 16 | 32 | 16384 | 32768 | 65536 | 131072 | 524288 | 2097152 | 4194304 | 8388608 | 16777216 | 134217728 | 268435456;
constexpr int WV_RESERVE = 32;  // packets a wave reserves per atomic on the chunk's packet counter

// per-wave LDS, structure of arrays indexed by lane
struct WaveShared {
    double nu[64], rcp_nu[64], comov_nu[64], chi[64], rcp_chi[64], tau_event[64], d_cont0[64];
    double d_boundary[64];  // in: boundary distance of the prepared trace; out: distance of the event found
    int res_info[64], res_line[64];
    unsigned rng_a[64], rng_b[64];  // lazy MT19937 seeding: init_genrand words mt[k] and mt[k+397] of the next block to regenerate
    // group sweeps only (the lane-sweep instantiations do not allocate the rest: 16 instead of 15 waves fit a CU's LDS)
    int cursor[64], rowfast[64];  // first line of the trace; shell * n_lines | exact-division fast path << 31
    int queue[64];                // lanes whose prepared trace waits for a worker group
};
constexpr size_t WAVE_SHARED_LS_BYTES = sizeof(WaveShared) - 3 * 64 * sizeof(int);
static_assert(offsetof(WaveShared, queue) + sizeof(int) * 64 == sizeof(WaveShared) && offsetof(WaveShared, cursor) + 3 * 64 * sizeof(int) == sizeof(WaveShared),
              "cursor | rowfast | queue are the last 768 bytes: the lane-sweep instantiations leave them out, the pooled volleys' item list lies over them");
struct WaveSharedFull {  // only read by the full-relativity sweep
    double r[64], mu[64];
};

template <bool FULL, bool VPK, bool LS = false, bool SL = false>
__host__ __device__ constexpr size_t wave_kernel_lds_bytes(int n_shells)
{
    return (LS ? WAVE_SHARED_LS_BYTES : sizeof(WaveShared)) + (FULL ? sizeof(WaveSharedFull) : 0) + (size_t)(VPK ? WV_RING_VPK : WV_RING) * 64 * sizeof(double) +
           (size_t)(VPK ? 6 : 5) * (size_t)n_shells * sizeof(double) +  // J, nu_bar, r_inner, r_outer, n_e (+ the tau row sums of the v-packet screening)
           (SL ? (size_t)n_shells * 2 * sizeof(unsigned) : 0);          // the open log chunk of every shell and its fill (shell-sorted log)
}

// result of one (possibly speculative) v-packet trace, handed from the worker lane to the owner lane through global scratch
struct __attribute__((aligned(16))) VpResult {
    double nu, energy, mu0;
    int used, visits, err, pad;
};
// A v-packet a worker lane was still tracing when the wave left its volley phase (cut-off with carry-over, WaveCold::vp_carry_min_active):
// parked here over the event phase of the next pass, picked up again by the same lane in that pass's volley phase.
struct __attribute__((aligned(16))) VpPark {
    double r, mu, nu, energy, tau, mu0, rcp_nu, margin, v0_r, v0_energy;
    int shell, next_line, owner, item, q, used, avail, head, v0_shell, v0_line;
    unsigned visits;
    int flags;  // 1: exact-division fast path, 2: screening
};
constexpr int VP_ROUND = 6;  // v-packets of one packet per round of a pooled volley (5 and 8 were measured: no difference)
static_assert(2 * VP_ROUND + 3 <= 16, "a round's mu and roulette draws, plus the 4 doubles of the refill that completes them, must fit WV_RING_VPK");

// Volley queue (variant 4).  On fine grids a v-packet crosses tens of shells and a volley's v-packets differ widely in length,
// so tracing them inside the wave that owns the packets leaves most lanes idle (23 of 64 busy on the 100-shell shape).  With the
// queue the propagation kernel only REQUESTS a round of v-packets -- the packet at the interaction, the look-ahead draws of its
// stream, the roulette predictor -- and suspends the lane; vpacket_trace_kernel, launched after it, traces the v-packets of all
// requests of the grid, one lane per v-packet, every lane pulling its next v-packet from the global list the moment it
// finishes one; the next launch of the propagation kernel commits the results in the reference's order (same validation of
// the predicted draw positions as the pooled volleys) and goes on.  The arithmetic of a v-packet is vp_shell_step()'s either way.
struct __attribute__((aligned(16))) VolleyRequest {
    double r, mu, nu, energy;          // the parent packet at the interaction
    int shell, next_line, vdone, cnt;  // v-packets of the volley already committed; draws[] holds cnt look-ahead doubles
    unsigned pred_bits;
    int pad0, pad1, pad2;
    double draws[WV_RING_VPK];         // the parent's stream from its current position
};
constexpr int VQ_RESERVE = 256;  // items a tracer wave reserves per atomic

// Kernel arguments.  Only what the sweep loop touches is passed by value (-> SGPRs); everything the event phase needs is
// read through `cold` (a device copy) at the top of every pass, so that it does not occupy scalar registers -- and, once
// those run out, VGPR lanes -- during the sweeps.
struct WaveHot {
    const double *nu_line, *tau_t;
    int n_lines, n_shells, disable_line_scattering, debug_flags;
    double t_exp;  // (c t and its reciprocal, only read by the prologue of a trace, come from the cold block: what is passed by value here lives in
                   // SGPRs through the sweep loop, and every one too many is moved through a VGPR lane there: 196 -> 96 lane moves)
    const int2 *line_block;           // lane sweep: macro-atom block of a line (null unless line_interaction_type != 0), requested as soon as a line stops the trace
    int ls_min_active, ls_max_steps;  // lane sweep: leave the sweep phase once this few lanes are still sweeping / after this many steps
    int walk_min_active;              // compact macro-atom walk: carry the walks over once this few lanes are still walking (-1: never)
    int vq_min_active;                // volley queue: end the launch once this few lanes can still go on while others wait for the tracer
};
struct LaunchRec;
// Suspended state of a lane / of a wave: a propagate call is a sequence of launches ("epochs") of the same grid over ONE
// packet supply.  A wave whose region of the line-visit log is full stores its lanes here and exits; the estimator passes
// consume the log while the next epoch -- which writes the other log buffer -- resumes every wave where it stopped.
// The lanes of a wave therefore never run dry between epochs: the drain of the longest-lived packets (0.26 s per launch on
// the macroatom shape, one live lane per wave) is paid once per call instead of once per log-bounded chunk.
struct __attribute__((aligned(16))) LaneSave {
    double r, mu, nu, energy, dop;
    double s_tau, s_tau_event, s_kp, s_xb;  // lane sweep in progress
    double d_cont0, d_boundary;             // sh.d_cont0 / sh.d_boundary (parameters of the running sweep or distance found)
    double ring[WV_RING_VPK];               // look-ahead doubles of the packet's MT19937 stream
    int shell, next_line, status, state, pkt, pflags, r_gpos, r_head, r_cnt;
    int trk_count, trk_boundary, flags;     // flags: 1 trk_any, 2 s_active, 4 s_fast
    int s_line;
    unsigned s_row;
    int res_info, res_line, pre_blk_x, pre_blk_y;
    unsigned rng_a, rng_b;
    int vseq;
    unsigned pred_bits;
    int vdone, pad_v0, pad_v1, pad_v2;  // volley queue: v-packets of the running volley committed so far (state WS_VOLLEY)
    // a carried-over macro-atom walk (state WS_WALK) and the interaction it belongs to: sh.chi | sh.rcp_chi | sh.nu | sh.rcp_nu | sh.comov_nu
    double walk_inv_new, walk_block, trk_nu, trk_mu, trk_energy;
    double walk_event, pad_w;  // sh.tau_event: the number a carried walk looks up again (WALK_REDO)
};
struct WaveSave {
    long long res_next, res_end;
    int exhausted, done;
    int log_chunk, log_gen;  // volley queue (log_continue): the chunk of the line-visit log the wave was appending to, and the pool's generation
    // volley queue: the work counters of the wave's earlier launches (a call of thousands of launches would otherwise send
    // thousands x waves x 7 atomics to the same seven words: ~1 ms per launch)
    unsigned long long cnt[7];
};
struct WaveCold {
    GroupArgs P;
    // A copy of the full problem description, NOT read through P.cold: loads through a pointer that was itself loaded from
    // memory cannot use the scalar cache (the kernel's own stores might alias them), so every `cold->field` access would be
    // a vector-memory round trip in the middle of the event phase; `W` is a restrict-qualified kernel argument.
    DeviceProblem D;
    EstimatorLog log;
    uint32_t *seeded_states;
    long long chunk_first, chunk_count;
    const LaunchRec *launch;  // [chunk_count] prepared packets (launch_prep_kernel)
    VpResult *vp_scratch;  // [waves][64 * VP_ROUND]
    VpPark *vp_park;       // [waves][64]; null: no carry-over
    int vp_carry_min_active, vp_pad;  // pooled volleys: leave the volley phase once nothing waits and this few lanes still trace (0: never)
    // epochs (see LaneSave): where the lanes / waves of this grid are suspended; resume = this launch continues them
    LaneSave *save;      // [waves * 64]
    WaveSave *wsave;     // [waves]
    int resume;
    // drain_split: a wave whose packet supply has run out suspends ONCE (at the top of its next pass) although its log region is
    // not full: the launch that follows drains the call's longest-lived packets -- a mostly idle chip for ~0.2 s on the macroatom
    // shape -- while the estimator passes consume everything logged so far on the second stream, instead of after the drain
    int drain_split;
    unsigned *suspended;  // [0] number of waves this launch suspended (0: the call is complete); [1] those whose log region is full; [2] those that split off their drain
    // volley queue (null / 0 unless variant 4): requests [waves * 64], items (slot << 3 | v-packet of the round), counters
    // {items, next item of the tracer}; log_continue: a resumed wave goes on appending to its log region (region_count)
    VolleyRequest *vq_req;
    unsigned *vq_items;
    unsigned *vq_count;
    int log_continue;
    int log_gen;       // generation of the log's chunk pool (the host bumps it whenever it resets the pool): a chunk held from an earlier one is gone
    double *vq_jsave;  // [waves][2 * n_shells]: the waves' J / nu_bar partial sums between the launches of a volley-queue call
};

__global__ void __launch_bounds__(64) late_list_kernel(const LaneSave *__restrict__ save, const WaveSave *__restrict__ wsave, int waves, long long first, long long end,
                                                       unsigned *__restrict__ late, unsigned *__restrict__ late_count, unsigned capacity)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= waves) return;
    const WaveSave ws = wsave[w];
    if (ws.done) return;  // (a finished wave holds nothing; its lanes' records are stale)
    const LaneSave &v = save[(size_t)w * 64 + lane];
    // (a packet in front of `first` that is still in flight was put on the list at an earlier boundary)
    const bool live = v.state != WS_NEED_PACKET && v.state != WS_DONE && (long long)v.pkt >= first && (long long)v.pkt < end;
    const long long r0 = ws.res_next > first ? ws.res_next : first, r1 = ws.res_end < end ? ws.res_end : end;
    const int n_res = r1 > r0 ? (int)(r1 - r0) : 0;  // reserved, not started (<= 64: a wave reserves max(32, lanes in need) at a time)
    const unsigned long long lm = __ballot(live);
    const unsigned n_live = (unsigned)__popcll(lm);
    unsigned base = 0;
    if (lane == 0 && n_live + (unsigned)n_res > 0) base = atomicAdd(late_count, n_live + (unsigned)n_res);
    base = (unsigned)__shfl((int)base, 0);
    if (live) {
        const unsigned pos = base + (unsigned)__popcll(lm & ((1ull << lane) - 1ull));
        if (pos < capacity) late[pos] = (unsigned)v.pkt;
    }
    for (int k = lane; k < n_res; k += 64) {
        const unsigned pos = base + n_live + (unsigned)k;
        if (pos < capacity) late[pos] = (unsigned)(r0 + k);
    }
}

// ---- drain compaction.  Once the packet supply has run out the lanes of a wave fall idle one by one while the wave keeps its place on the chip to the end of its
// longest packet: a chip full of resident waves with a handful of live lanes each, and no room beside them for the estimator passes of the epoch that has just ended.
// With WaveCold::drain_split = T the waves suspend when T or fewer of their lanes are left; the kernels below then pack the live lanes of all suspended waves into
// waves of 64 -- a lane's whole state is its LaneSave record and its MT19937 state buffer, neither depends on the place in the grid -- and the rest of the call runs
// as a launch of a quarter of the waves (or fewer) beside the passes.
// census: [0] live lanes, [1] packets reserved but not started, [2] waves not done, [3] lanes waiting for a packet (must be 0 where [1] is: nothing left to hand out)
__global__ void __launch_bounds__(64) drain_census_kernel(const LaneSave *__restrict__ save, const WaveSave *__restrict__ wsave, int waves, long long n_packets,
                                                          unsigned *__restrict__ out)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= waves) return;
    const WaveSave ws = wsave[w];
    if (ws.done) return;
    const int state = save[(size_t)w * 64 + lane].state;
    const unsigned n_live = (unsigned)__popcll(__ballot(state != WS_NEED_PACKET && state != WS_DONE));
    const unsigned n_need = (unsigned)__popcll(__ballot(state == WS_NEED_PACKET));
    const long long r1 = ws.res_end < n_packets ? ws.res_end : n_packets;
    if (lane == 0) {
        if (n_live) atomicAdd(out, n_live);
        if (r1 > ws.res_next) atomicAdd(out + 1, (unsigned)(r1 - ws.res_next));
        atomicAdd(out + 2, 1u);
        if (n_need) atomicAdd(out + 3, n_need);
    }
}
// live lanes of wave w -> consecutive places of the packed grid (the order is that of the atomics: a packet's results do not depend on its place)
// (`density`: live lanes per packed wave, <= 64 -- a pass takes longer the more of a wave's lanes are live, and the drain is the chain of its longest packet)
__global__ void __launch_bounds__(64) drain_compact_kernel(const LaneSave *__restrict__ save, const WaveSave *__restrict__ wsave, const uint32_t *__restrict__ states, int waves,
                                                           LaneSave *__restrict__ dst_save, uint32_t *__restrict__ dst_states, unsigned *__restrict__ counter, unsigned density)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= waves) return;
    const WaveSave ws = wsave[w];
    if (ws.done) return;
    const int state = save[(size_t)w * 64 + lane].state;
    const bool live = state != WS_NEED_PACKET && state != WS_DONE;
    const unsigned long long lm = __ballot(live);
    if (!lm) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned)__popcll(lm));
    base = (unsigned)__shfl((int)base, 0);
    auto place = [density](unsigned g) { return (size_t)(g / density) * 64 + (g % density); };  // g-th live lane of the grid -> its wave and lane
    if (live) dst_save[place(base + (unsigned)__popcll(lm & ((1ull << lane) - 1ull)))] = save[(size_t)w * 64 + lane];
    // the MT19937 state buffers of the live lanes, one after the other, 64 words at a time
    unsigned k = 0;
    for (unsigned long long m = lm; m; m &= m - 1ull, ++k) {
        const int src_lane = __ffsll((long long)m) - 1;
        const uint32_t *from = states + ((size_t)w * 64 + (size_t)src_lane) * WV_STATE_STRIDE;
        uint32_t *to = dst_states + place(base + k) * WV_STATE_STRIDE;
        for (int i = lane; i < WV_STATE_STRIDE; i += 64) to[i] = from[i];
    }
}
// the packed grid's wave records, and the idle lanes behind the last live one
__global__ void __launch_bounds__(256) drain_compact_finish_kernel(LaneSave *__restrict__ dst_save, WaveSave *__restrict__ dst_wsave, const unsigned *__restrict__ counter,
                                                                  long long n_packets, unsigned density)
{
    const unsigned total = *counter, waves = (total + density - 1u) / density;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= waves * 64u) return;
    if ((i & 63u) >= density || (i >> 6) * density + (i & 63u) >= total) {
        LaneSave idle{};
        idle.state = WS_DONE;
        dst_save[i] = idle;
    }
    if ((i & 63u) == 0u) {
        WaveSave ws;
        ws.res_next = ws.res_end = n_packets; ws.exhausted = 1; ws.done = 0; ws.log_chunk = -1; ws.log_gen = 0;
        for (int k = 0; k < 7; ++k) ws.cnt[k] = 0ull;
        dst_wsave[i >> 6] = ws;
    }
}

// One worker slot of a group: the trace it is sweeping (group-uniform values) and this lane's line of the current chunk.
struct SweepSlot {
    int owner;  // lane (in this wave) of the packet being traced, -1: idle
    double nu, rcp_nu, comov_nu, chi, rcp_chi, tau_event, d_boundary, r, mu;
    double tau_carry, d_cont_carry;
    double nu_line, tau_line;
    int cur0;
    unsigned row;
    bool fast;
};

// One G-line step of the line sweep of trace_packet (modes/homologous_rad_packet_transport.py:100-172); the loop body of
// sweep_lines() in propagate_group.hpp with the loop turned inside out and without the estimator updates.  On a stop
// (or when the list is exhausted) the result is handed to the owner lane through LDS and the slot becomes idle.
template <bool FULL, int G, bool FAST>
__device__ __forceinline__ void sweep_step(const WaveHot &P, SweepSlot &s, const int j, WaveShared &sh, unsigned long long &visits)
{
    const int L = P.n_lines;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : ((G == 8) ? 0xffull : 0xfull);
    if (s.cur0 < L) {
        const int line = s.cur0 + j;
        const bool in_range = line < L;
        // prefetch the next chunk
        const int nline = line + G;
        const bool nin = nline < L;
        const double nu_next = nin ? P.nu_line[(unsigned)nline] : 0.0;
        const double tau_next = nin ? P.tau_t[s.row + (unsigned)nline] : 0.0;

        const double nu_line = s.nu_line;
        const double tau_incl = serial_prefix<G>(s.tau_carry, s.tau_line, j);
        const double tau_prev = group_shr1<G>(s.tau_carry, tau_incl, j);
        const double d_cont = (j == 0) ? s.d_cont_carry : exact_div<FAST>(s.tau_event - tau_prev, s.chi, s.rcp_chi);
        const bool is_last = line == L - 1;
        const double nu_diff = s.comov_nu - nu_line;
        const double q = exact_div<FAST>(nu_diff, s.nu, s.rcp_nu);
        const bool close = fabs(q) < CLOSE_LINE_THRESHOLD;
        const bool err = in_range && !is_last && !close && !(nu_diff >= 0);
        double d_far;
        if (FULL) d_far = distance_line_full_relativity(nu_line, s.nu, P.t_exp, s.r, s.mu);
        else d_far = q * C_LIGHT * P.t_exp;
        const double d_trace = is_last ? MISS_DISTANCE : (close ? 0.0 : d_far);
        const double tau_combined = tau_incl + s.chi * d_trace;
        double dmin = d_trace;
        if (s.d_boundary < dmin) dmin = s.d_boundary;
        if (d_cont < dmin) dmin = d_cont;
        const bool ok = in_range && !err;
        const bool stop_b = ok && d_trace != 0 && dmin == s.d_boundary;
        const bool stop_e = ok && d_trace != 0 && !stop_b && dmin == d_cont;
        const bool stop_l = ok && !stop_b && !stop_e && tau_combined > s.tau_event && !P.disable_line_scattering;
        const bool stop = stop_b || stop_e || stop_l || (in_range && err);
        const unsigned stop_mask = (unsigned)((__ballot(stop) >> gshift) & GMASK);
        if (stop_mask) {
            const int first = __builtin_ctz(stop_mask);
            visits += (unsigned long long)(first + 1);
            if (j == first) {  // the stopping lane hands the result to the owner lane
                sh.d_boundary[s.owner] = stop_b ? s.d_boundary : (stop_e ? d_cont : d_trace);
                sh.res_line[s.owner] = line;
                sh.res_info[s.owner] = stop_b ? 1 : (stop_e ? 2 : (stop_l ? 3 : 4));
            }
            s.owner = -1;
        } else {
            const int n_in = min(G, L - s.cur0);
            visits += (unsigned long long)n_in;
            s.tau_carry = gbcast<G>(tau_incl, n_in - 1);
            s.d_cont_carry = exact_div<FAST>(s.tau_event - s.tau_carry, s.chi, s.rcp_chi);
            s.cur0 += G;
            s.nu_line = nu_next;
            s.tau_line = tau_next;
        }
    } else {
        // for-else (lines 157-172): the line list is exhausted; next_line_id is left untouched (bit 3)
        if (j == 0) {
            const bool cont = s.d_cont_carry < s.d_boundary;
            sh.d_boundary[s.owner] = cont ? s.d_cont_carry : s.d_boundary;
            sh.res_line[s.owner] = 0;
            sh.res_info[s.owner] = (cont ? 2 : 1) | 8;
        }
        s.owner = -1;
    }
}


// ---- lane sweep (LS instantiations, partial relativity): every lane sweeps the line list of ITS packet, eight lines per step.
//
// The reference's per-line work (distance to the line, three-way minimum, optical-depth test) only matters at the line
// where the trace stops; for all lines before it, it is enough to PROVE that the reference's tests come out negative.
// With X = comov_nu - nu_line >= 0, K' >= chi C t / nu (rounded up by 2^-40 relative) and x = RN(K' X):
//     RN(chi d_trace) <= x                       (d_trace = RN(RN(RN(X / nu) C) t) <= (X C t / nu)(1 + 3u), or 0 for a close line)
//     X  <  X_b = (d_boundary nu / (C t))(1 - 2^-40)   =>  d_trace < d_boundary
//     x  <  RN(tau_event - tau_prev)             =>  d_trace < d_continuum = RN(RN(tau_event - tau_prev) / chi)
//     RN(tau_incl + x) <= tau_event              =>  !(tau_combined > tau_event)          (rounding is monotone)
// so the line cannot stop the trace and only tau_incl = tau_prev + tau_line (the reference's serial sum) is carried on:
// 5 flops and 4 compares per line instead of two divisions and the full predicate.  The first line that fails one of the
// bounds (and the last line of the list, and every line of a trace whose operands are outside mid_range) is evaluated
// with the reference's own arithmetic by lane_exact_line() in the same step.
__device__ __forceinline__ int lane_exact_line(const WaveHot &P, int line, double nu_line, double tau_line, double tau_prev, double nu,
                                               double comov_nu, double chi, double tau_event, double d_boundary, double &distance)
{   // trace_packet's loop body for one line (modes/homologous_rad_packet_transport.py:100-156), as in sweep_step(); the two
    // quotients are plain divisions here (exact_div<true> returns the same correctly rounded values)
    const double tau_incl = tau_prev + tau_line;
    const double d_cont = (tau_event - tau_prev) / chi;
    const bool is_last = line == P.n_lines - 1;
    const double nu_diff = comov_nu - nu_line;
    const double q = nu_diff / nu;
    const bool close = fabs(q) < CLOSE_LINE_THRESHOLD;
    const bool err = !is_last && !close && !(nu_diff >= 0);
    const double d_far = q * C_LIGHT * P.t_exp;
    const double d_trace = is_last ? MISS_DISTANCE : (close ? 0.0 : d_far);
    const double tau_combined = tau_incl + chi * d_trace;
    double dmin = d_trace;
    if (d_boundary < dmin) dmin = d_boundary;
    if (d_cont < dmin) dmin = d_cont;
    const bool stop_b = !err && d_trace != 0 && dmin == d_boundary;
    const bool stop_e = !err && d_trace != 0 && !stop_b && dmin == d_cont;
    const bool stop_l = !err && !stop_b && !stop_e && tau_combined > tau_event && !P.disable_line_scattering;
    distance = stop_b ? d_boundary : (stop_e ? d_cont : d_trace);
    return stop_b ? 1 : (stop_e ? 2 : (stop_l ? 3 : (err ? 4 : 0)));
}
constexpr int LS_CHUNK = 8;  // lines per lane and step (the line list and the tau table carry this much slack at the end)
#ifndef TMC_STRAIGHT_A
#define TMC_STRAIGHT_A 0  // (experiment switch: the straight-line sweep form in the sixteen-wave instantiation, too; profiles/r06_lines_per_step.txt)
#endif
#ifndef TMC_LS_CHUNK_NT
#define TMC_LS_CHUNK_NT 10
#endif
// ... of the sixteen-wave instantiation on the interleaved table (its slack: 32 entries).  Ten: with the lean proof the 128-VGPR budget holds two more lines (14 instead
// of 10 spilled VGPRs), a 36-line trace is 4.1 instead of 5 dependent steps: propagation launches -2.4 % on configs[2] heavy (2 748 / 2 684 -> 2 652 / 2 651 ms per 1e8
// packets), nothing either way on the uniform levels, the 1.25e7-packet call and configs[1]; twelve (29 spilled VGPRs): +3 ... +7 % (profiles/r06_lines_per_step.txt)
constexpr int LS_CHUNK_NT = TMC_LS_CHUNK_NT;

// Running sums of the transition probabilities, block by block (macro_atom.py:87-97: `probability += transition_probabilities[i, shell]`
// from block_start): one thread per (block, shell) repeats the reference's additions once per opacity state, so that a
// jump becomes a search for the first running sum that exceeds the drawn number.  `negative` is raised if a probability is
// negative (the sums would not be monotone; the host then keeps such problems off the searching kernel).
__global__ void __launch_bounds__(256) macro_cumulative_kernel(const double *__restrict__ prob_t, double *__restrict__ cum_t,
                                                                const int *__restrict__ block_edge, int n_blocks, long long n_trans,
                                                                int n_shells, int *negative)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_blocks * n_shells) return;
    const int b = (int)(i % n_blocks), s = (int)(i / n_blocks);
    const double *p = prob_t + (long long)s * n_trans;
    double *c = cum_t + (long long)s * n_trans;
    double carry = 0.0;
    bool neg = false;
    const int b0 = block_edge[b], b1 = block_edge[b + 1];
    for (int k0 = b0; k0 < b1; k0 += 8) {  // (eight probabilities requested together; the additions stay the reference's, one after the other)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = k0 + q < b1 ? p[k0 + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (k0 + q < b1) {
                neg |= !(v[q] >= 0.0);
                carry += v[q];
                c[k0 + q] = carry;
            }
    }
    if (neg) atomicOr(negative, 1);
}

// Last-interaction tracker (packets/trackers/tracker_last_interaction.py:8-254) as the wave kernel keeps it: one 64-byte record
// per packet, rewritten at every interaction with four 16-byte stores (one write request; nine 8-byte stores into nine
// arrays were nine requests, a quarter of all memory requests of the macroatom workload) and unpacked into the boundary's
// arrays by tracker_unpack_kernel after the propagation.  The fields the reference fills at the end of a packet (nu, energy,
// after_nu, after_energy) equal the packet's outputs.
struct __attribute__((aligned(16))) TrackerRecord {
    double before_nu, before_mu, before_energy, radius, after_mu;
    int shell, type, absorb, emit, count, valid;
};
static_assert(sizeof(TrackerRecord) == 64, "TrackerRecord is one 64-byte request");

// (`first`, `count`: the packet range [first, first + count) -- the whole call, or the part of it whose results are streamed to the host while the
// propagation is still running; `index` non-null: the packets index[0 .. count) instead, the late finishers of streamed ranges)
__global__ void __launch_bounds__(256) tracker_unpack_kernel(DeviceProblem D, long long first, long long count, const unsigned *__restrict__ index = nullptr)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const long long i = index ? (long long)index[j] : first + j;
    const double e = D.out_e[i];
    if (e == -99.0) return;  // the packet ended with an error: its tracker fields stay as they were
    const TrackerRecord r = reinterpret_cast<const TrackerRecord *>(D.li_rec)[i];
    const bool any = r.valid != 0 && r.count > 0;
    const double nan = __builtin_nan("");
    const double nu = D.out_nu[i], en = fabs(e);
    D.li_radius[i] = any ? r.radius : nan;
    D.li_nu[i] = any ? nu : nan; D.li_energy[i] = any ? en : nan;
    D.li_before_nu[i] = any ? r.before_nu : nan; D.li_before_mu[i] = any ? r.before_mu : nan; D.li_before_energy[i] = any ? r.before_energy : nan;
    D.li_after_nu[i] = any ? nu : nan; D.li_after_mu[i] = any ? r.after_mu : nan; D.li_after_energy[i] = any ? en : nan;
    D.li_shell_id[i] = any ? r.shell : -1; D.li_interaction_type[i] = any ? r.type : -1;
    D.li_line_absorb_id[i] = any ? r.absorb : -1; D.li_line_emit_id[i] = any ? r.emit : -1;
    D.li_interactions_count[i] = any ? r.count : 0;
}

// ---- result streaming (round 6): the packets that had been handed out but not finished when an epoch ended -- the live lanes of the suspended waves and the
// packets a wave has reserved and not started -- are the ones whose outputs the host may copy too early while the next epoch runs; their indices are collected
// here (one atomic per wave) and their results are sent again when the call is over.  One 64-thread workgroup per wave of the propagation grid.
// (late_list_kernel: below, behind LaneSave / WaveSave.)
// dst[j] = src[index[j]] (8-byte elements): the late finishers' values of one per-packet array, compacted for the copy to the host
__global__ void __launch_bounds__(256) gather64_kernel(const unsigned long long *__restrict__ src, const unsigned *__restrict__ index, long long n,
                                                       unsigned long long *__restrict__ dst)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[index[j]];
}

// What a lane needs to start a packet, prepared by launch_prep_kernel for the whole chunk (one record per packet, 48 bytes): the
// packet in the lab frame (set_packet_props_{partial,full}_relativity, classic/packet_propagation.py:254-318), its first line
// (initialize_line_id, packets/radiative_packet.py:96-110), its seed and word 397 of its init_genrand sequence (the only
// part of the MT19937 start state that is ever precomputed, see refill()).  Fetching a packet is then one round trip
// instead of a chain of four (inputs, bucket index, line scan).
struct __attribute__((aligned(16))) LaunchRec {
    double r, mu, nu, energy;
    uint32_t seed, checkpoint;
    int line0, pad;
};
struct LaunchPrepArgs {
    const double *r0, *mu0, *nu0, *e0, *nu_line;
    const uint32_t *seeds;
    const int *bucket_first;
    int bucket_shift, bucket_n, n_lines;
    long long bucket_kmin;
    double t_exp;
    LaunchRec *out;
    long long first, count;
};
template <bool FULL>
__global__ void __launch_bounds__(256) launch_prep_kernel(LaunchPrepArgs a)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.count) return;
    const long long i = a.first + j;
    LaunchRec rec;
    rec.seed = a.seeds[i];
    uint32_t x = rec.seed;
    for (int k = 1; k <= 397; ++k) x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
    rec.checkpoint = x;
    double r = a.r0[i], mu = a.mu0[i], nu = a.nu0[i], energy = a.e0[i];
    const double t = a.t_exp;
    {
        const double velocity = r / t;
        const double inv = inverse_doppler_factor<FULL>(velocity, mu);
        if (FULL) {
            const double beta = velocity / C_LIGHT;
            nu *= inv; energy *= inv;
            mu = (mu + beta) / (1 + beta * mu);
        } else { nu *= inv; energy *= inv; }
    }
    const int L = a.n_lines;
    {
        const double velocity = r / t;
        const double comov_nu = nu * doppler_factor<FULL>(velocity, mu);
        int lo;
        const long long kk = (long long)((unsigned long long)__double_as_longlong(comov_nu > 0.0 ? comov_nu : 0.0) >> a.bucket_shift) - a.bucket_kmin;
        if (kk >= a.bucket_n) lo = 0;
        else if (kk < 0) lo = L;
        else {
            lo = a.bucket_first[kk];
            const int hi = kk > 0 ? a.bucket_first[kk - 1] : L;
            while (lo < hi && a.nu_line[(unsigned)lo] >= comov_nu) ++lo;
        }
        if (lo == L) lo -= 1;
        rec.line0 = lo;
    }
    rec.r = r; rec.mu = mu; rec.nu = nu; rec.energy = energy; rec.pad = 0;
    a.out[j] = rec;
}

// ---- v-packets (packets/virtual_packet.py:82-386), lane-per-packet: every lane traces the v-packets of ITS packet one after
// the other, drawing from its own stream in the reference's order -- so, unlike the group kernel, nothing has to be
// predicted or re-traced.  The per-shell work is vp_trace()'s of the group kernel: the stopping line is located through
// the frequency-bucket index and pinned with the reference's own predicate, the optical depths are summed in the
// reference's order.  The volley is a flat per-lane state machine (one shell crossing per pass of the wave-level loop; a
// lane that finishes a v-packet starts its next one in the following pass), so lanes never wait for the longest
// v-packet of a round.  `draws_left` bounds the Russian-roulette draws one v-packet may take from the lane's LDS ring.
struct VpState {
    double r, mu, nu, energy, tau, mu0;
    int shell, next_line;
};

// one shell crossing of trace_vpacket (:82-244): returns 1 when the v-packet has left the grid / died, 0 to go on, < 0 error.
// Written branch-light so that the lanes of a wave (each on a different v-packet) stay converged: the stopping line is
// pinned with a fixed number of predicate evaluations around the frequency-bucket guess (a loop only if that was not
// enough), and the optical depths are added in wave-uniform chunks of 8 with exact no-op adds (+0.0) in the lanes that
// have fewer lines.  rcp_nu = RN(1 / v.nu) serves the exact 3-instruction division (mc_device.hpp) of the resonance
// distances; v.nu is constant along a v-packet.
template <bool FULL, typename Draw>
__device__ __forceinline__ int vp_shell_step(const GroupArgs &P, Draw &&draw, int &draws_left, VpState &v, double rcp_nu, bool fast_nu,
                                             const double *__restrict__ geo /* LDS: r_inner | r_outer | n_e */, unsigned &vvisits)
{
    const int L = P.n_lines, S = P.n_shells;
    const double t = P.t_exp;
    int status = ST_IN_PROCESS;
    const unsigned row = (unsigned)v.shell * (unsigned)L;
    const int start = v.next_line;
    // The step is bound by the latency of its dependent, uncoalesced loads, so everything whose address is known now is
    // requested first: the line at `start` and -- speculatively -- the first eight optical depths of the sum.
    const int start_c = min(start, L - 1);
    const MC_G double *__restrict__ trow = glob(P.tau_t) + row + (unsigned)start_c;
    const MC_G double *__restrict__ nu_line_g = glob(P.nu_line);
    const double nl_start = nu_line_g[(unsigned)start_c];
    // (two optical depths per load instruction; past the end of the table there is slack, and what lies beyond the lines of the
    // sum is replaced by +0.0 before it is used)
    typedef double tau2 __attribute__((ext_vector_type(2), aligned(8)));
    double tv[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const tau2 w = *reinterpret_cast<const MC_G tau2 *>(trow + k);
        tv[k] = w.x; tv[k + 1] = w.y;
    }
    // trace_vpacket_within_shell (:82-175)
    double d_boundary;
    int delta;
    distance_boundary(v.r, v.mu, geo[v.shell], geo[S + v.shell], d_boundary, delta);
    const double chi_e = geo[2 * S + v.shell] * P.sigma_thomson;
    const double velocity = v.r / t;
    const double dop = doppler_factor<FULL>(velocity, v.mu);
    const double comov_nu = v.nu * dop;
    double chi_cont = chi_e;
    if (FULL) chi_cont *= dop;
    double tau_shell = chi_cont * d_boundary;
    // the frequency bucket of the shell boundary needs nothing that is still in flight: its lookup travels with the loads
    // above instead of forming a round trip of its own after them
    const double nu_thr = comov_nu - d_boundary * P.rcp_tc * v.nu;
    long long kk_b = (long long)((unsigned long long)__double_as_longlong(nu_thr > 0.0 ? nu_thr : 0.0) >> P.bucket_shift) - P.bucket_kmin;
    kk_b = kk_b < 0 ? 0 : (kk_b >= P.bucket_n ? P.bucket_n - 1 : kk_b);
    const int bucket_e = glob(P.bucket_first)[kk_b];
    // calculate_distance_line (calculate_distances.py:66-112) of line k (frequency nl) for this v-packet
    auto d_line_of = [&](int k, double nl) -> double {
        if (FULL) {
            double d;
            distance_line<FULL>(v.nu, v.r, v.mu, comov_nu, k == L - 1, nl, t, d);
            return d;
        }
        const double nu_diff = comov_nu - nl;
        const double q = (fast_nu && mid_range(nu_diff)) ? exact_div<true>(nu_diff, v.nu, rcp_nu) : nu_diff / v.nu;
        const double d = (fabs(q) < CLOSE_LINE_THRESHOLD) ? 0.0 : q * C_LIGHT * t;
        return (k == L - 1) ? MISS_DISTANCE : d;
    };
    int n_sum = 0;
    if (start < L) {
        double d_line;
        // the reference evaluates line `start` first and raises there if it lies blueward of the packet (lines further
        // down the sorted list can then not raise: their nu_diff is larger)
        if (!distance_line<FULL>(v.nu, v.r, v.mu, comov_nu, start == L - 1, nl_start, t, d_line)) return ERR_MONTECARLO;
        int e = start;
        if (!(d_boundary <= d_line)) {
            // first line after `start` whose resonance lies at or beyond the shell boundary (monotone along the list):
            // the frequency-bucket index gives a guess, a window of four lines around it is tested in one round trip
            e = max(bucket_e, start + 1);
            if (e > L - 1) e = L - 1;
            const int w0 = max(e - 1, start + 1);
            // (two 16-byte loads: the list ends in slack, and an index clamped to the last line does not look at its frequency)
            typedef double nu2 __attribute__((ext_vector_type(2), aligned(8)));
            const nu2 wa = *reinterpret_cast<const MC_G nu2 *>(nu_line_g + (unsigned)w0), wb = *reinterpret_cast<const MC_G nu2 *>(nu_line_g + (unsigned)w0 + 2);
            const double wn[4] = {wa.x, wa.y, wb.x, wb.y};
            bool sw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = d_boundary <= d_line_of(min(w0 + i, L - 1), wn[i]);
            bool resolved = false, stops = false;
            if (sw[0]) {
                if (w0 == start + 1) { e = w0; stops = true; resolved = true; }
                else e = w0;  // the first stopping line lies before the window
            } else if (sw[1]) { e = min(w0 + 1, L - 1); stops = true; resolved = true; }
            else if (sw[2]) { e = min(w0 + 2, L - 1); stops = true; resolved = true; }
            else if (sw[3]) { e = min(w0 + 3, L - 1); stops = true; resolved = true; }
            else e = min(w0 + 3, L - 1);  // beyond the window
            if (!resolved) {  // the bucket guess was further off: the reference's walk, four lines per round trip
                e = vp_walk_to_stop(nu_line_g, L, start, w0, sw[0], [&](int k, double nl) { return d_boundary <= d_line_of(k, nl); });
                stops = e < L;
            }
            if (!stops) e = L;  // (the reference then sums every line)
        }
        n_sum = min(e, L) - start;
        vvisits += (unsigned)((e < L) ? (e - start + 1) : (L - start));
        v.next_line = e;
    }
    // serial-order sum of tau over [start, start + n_sum): wave-uniform chunks of 8 (the first one was requested at the top),
    // +0.0 where a lane has no line left
#pragma unroll
    for (int k = 0; k < 8; ++k) tau_shell += (k < n_sum) ? tv[k] : 0.0;
    for (int base = 8; __ballot(base < n_sum); base += 8) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const int o = base + k;
            const tau2 z = {0.0, 0.0};
            const tau2 w = (o < n_sum) ? *reinterpret_cast<const MC_G tau2 *>(trow + o) : z;
            tv[k] = w.x; tv[k + 1] = (o + 1 < n_sum) ? w.y : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tau_shell += tv[k];
    }
    // trace_vpacket (:179-244)
    v.tau += tau_shell;
    cross_shell(v.shell, status, delta, P.n_shells);
    if (v.tau > P.tau_russian) {
        if (draws_left <= 0) return ERR_UNSUPPORTED;
        --draws_left;
        const double ev = draw();
        if (ev > P.survival_probability) {
            v.energy = 0.0;
            status = ST_EMITTED;
        } else {
            v.energy = v.energy / P.survival_probability * mcm::exp(-v.tau);
            v.tau = 0.0;
        }
    }
    const double new_r = sqrt(v.r * v.r + d_boundary * d_boundary + 2.0 * v.r * d_boundary * v.mu);
    v.mu = (v.mu * v.r + d_boundary) / new_r;
    v.r = new_r;
    return status == ST_EMITTED ? 1 : 0;
}

// One shell crossing of the SCREENING trace (tau_prefix.hpp): the crossing's optical depth from the prefix sums of the shell's row
// instead of its lines -- two round trips (line at `start`, prefix at `start`, frequency bucket of the boundary | four-line window
// of frequencies and prefix sums around the bucket guess).  v.tau / margin accumulate the approximate depth and its rigorous error
// bound.  Returns 1: the v-packet is certainly dropped by the roulette (one draw consumed, energy 0); 0: go on; 2: undecided here
// (leaves the grid alive, within the margin of the threshold, or drew exactly 0.0): the caller restarts it with vp_shell_step.
template <bool FULL, typename Draw>
__device__ __forceinline__ int vp_screen_step(const GroupArgs &P, Draw &&draw, int &draws_left, VpState &v, double &margin, double rcp_nu, bool fast_nu,
                                              const double *__restrict__ geo /* LDS: r_inner | r_outer | n_e | tau row sums */, unsigned &vvisits)
{
    const int L = P.n_lines, S = P.n_shells;
    const double t = P.t_exp;
    int status = ST_IN_PROCESS;
    const int start = v.next_line;
    const int start_c = min(start, L - 1);
    const MC_G double *__restrict__ prow = glob(P.tau_pfx) + (size_t)v.shell * (size_t)(L + 1);
    const MC_G double *__restrict__ nu_line_g = glob(P.nu_line);
    const double nl_start = nu_line_g[(unsigned)start_c];
    const double p_start = prow[(unsigned)min(start, L)];
    double d_boundary;
    int delta;
    distance_boundary(v.r, v.mu, geo[v.shell], geo[S + v.shell], d_boundary, delta);
    const double velocity = v.r / t;
    const double dop = doppler_factor<FULL>(velocity, v.mu);
    const double comov_nu = v.nu * dop;
    double chi_cont = geo[2 * S + v.shell] * P.sigma_thomson;
    if (FULL) chi_cont *= dop;
    const double tau_cont = chi_cont * d_boundary;
    const double nu_thr = comov_nu - d_boundary * P.rcp_tc * v.nu;
    long long kk_b = (long long)((unsigned long long)__double_as_longlong(nu_thr > 0.0 ? nu_thr : 0.0) >> P.bucket_shift) - P.bucket_kmin;
    kk_b = kk_b < 0 ? 0 : (kk_b >= P.bucket_n ? P.bucket_n - 1 : kk_b);
    const int bucket_e = glob(P.bucket_first)[kk_b];
    auto d_line_of = [&](int k, double nl) -> double {
        if (FULL) {
            double d;
            distance_line<FULL>(v.nu, v.r, v.mu, comov_nu, k == L - 1, nl, t, d);
            return d;
        }
        const double nu_diff = comov_nu - nl;
        const double q = (fast_nu && mid_range(nu_diff)) ? exact_div<true>(nu_diff, v.nu, rcp_nu) : nu_diff / v.nu;
        const double d = (fabs(q) < CLOSE_LINE_THRESHOLD) ? 0.0 : q * C_LIGHT * t;
        return (k == L - 1) ? MISS_DISTANCE : d;
    };
    double seg = 0.0;
    int n_sum = 0;
    if (start < L) {
        double d_line;
        if (!distance_line<FULL>(v.nu, v.r, v.mu, comov_nu, start == L - 1, nl_start, t, d_line)) return ERR_MONTECARLO;
        int e = start;
        if (!(d_boundary <= d_line)) {
            e = max(bucket_e, start + 1);
            if (e > L - 1) e = L - 1;
            const int w0 = max(e - 1, start + 1);
            typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
            const dbl2 wa = *reinterpret_cast<const MC_G dbl2 *>(nu_line_g + (unsigned)w0), wb = *reinterpret_cast<const MC_G dbl2 *>(nu_line_g + (unsigned)w0 + 2);
            const dbl2 pa = *reinterpret_cast<const MC_G dbl2 *>(prow + (unsigned)w0), pb = *reinterpret_cast<const MC_G dbl2 *>(prow + (unsigned)w0 + 2);
            const double wn[4] = {wa.x, wa.y, wb.x, wb.y}, wp[4] = {pa.x, pa.y, pb.x, pb.y};
            bool sw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[i] = d_boundary <= d_line_of(min(w0 + i, L - 1), wn[i]);
            int hit = -1;
            if (sw[0]) { if (w0 == start + 1) hit = 0; }
            else if (sw[1]) hit = 1;
            else if (sw[2]) hit = 2;
            else if (sw[3]) hit = 3;
            if (hit >= 0 && w0 + hit <= L - 1) {
                e = w0 + hit;
                double pe = wp[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) if (hit == i) pe = wp[i];
                seg = pe - p_start;
            } else {  // the bucket guess was further off: the reference's walk, four lines per round trip
                if (hit >= 0) e = L - 1;  // (the window ran past the list: its last line stops)
                else e = vp_walk_to_stop(nu_line_g, L, start, w0, sw[0], [&](int k, double nl) { return d_boundary <= d_line_of(k, nl); });
                seg = prow[(unsigned)min(e, L)] - p_start;
            }
        }
        n_sum = min(e, L) - start;
        vvisits += (unsigned)((e < L) ? (e - start + 1) : (L - start));
        v.next_line = e;
    }
    const double tau_shell = tau_cont + seg;
    v.tau += tau_shell;
    margin += 2.3e-16 * (geo[3 * S + v.shell] + (double)(n_sum + 4) * tau_shell + 2.0 * v.tau);
    cross_shell(v.shell, status, delta, S);
    if (v.tau - 2.0 * margin > P.tau_russian) {  // the reference's `tau_trace_combined > tau_russian` is certainly true
        if (draws_left <= 0) return ERR_UNSUPPORTED;
        --draws_left;
        const double ev = draw();
        if (!(ev > P.survival_probability)) return 2;  // (a draw of exactly 0.0)
        v.energy = 0.0;
        return 1;
    }
    if (!(v.tau + 2.0 * margin < P.tau_russian)) return 2;  // too close to call
    if (status == ST_EMITTED) return 2;                     // leaves the grid alive: its energy needs the reference's own sum
    const double new_r = sqrt(v.r * v.r + d_boundary * d_boundary + 2.0 * v.r * d_boundary * v.mu);
    v.mu = (v.mu * v.r + d_boundary) / new_r;
    v.r = new_r;
    return 0;
}

// XWALK: with the macro-atom walks on the fp64 running sums compiled in (the cooperative group scan and the per-lane search: what the
// wave kernel runs when the compact walk tables are not used -- debug flags 128 / 8192, cross-checks); the production instantiations
// leave them out: ~1 100 instructions and two inlined MT19937 refills less.
// WPE: waves per SIMD the instantiation is compiled for (its VGPR budget: 128 at 4, 168 at 3, 256 at 2).  The v-packet instantiations spill
// 178 VGPRs at 3; where the LDS of a wave (the per-shell arrays of a fine grid) allows no more than eight waves per CU anyway, the host
// launches the WPE = 2 instantiation: 239 VGPRs, nothing spilled, no scratch.
// NT (lane-sweep instantiations without v-packets only): the sweep reads the interleaved table nt_t[shell][line] = {nu_line, tau} (16 bytes per
// line, rows of `nt_stride` entries starting on 128-byte boundaries; H.tau_t points at it) instead of the line list and the shell's tau row:
// the eight 16-byte loads of a step then come from ONE run of 128 bytes instead of two runs of 64 bytes in two tables.  1: the run starts
// at the trace's current line (as the separate tables' chunks do); 2: the run is the aligned 128-byte line that holds the current line --
// a step never straddles two lines, the entries in front of the current line are skipped (the first step of a trace is shorter).
// SL (shell-sorted log, round 6; n_shells <= 64, no volley queue): a wave keeps one open chunk of the line-visit log PER SHELL (chunk and fill in LDS) and appends
// a trace's record to the chunk of its shell (one LDS atomic on the shell's fill), so that every chunk of the log holds records of ONE shell and the estimator passes
// start with the partition by bin: the pass that grouped the log by shell (a full read and write of the log: 21 of the 87 ms of passes per 2e9 records,
// profiles/r04_estimator_partition.txt) is gone.
template <bool FULL, bool TRACK, int G, bool VPK, bool LS = false, bool XWALK = true, int WPE = (VPK ? 3 : 4), int NT = 0, bool SL = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) propagate_wave_kernel(WaveHot H, const WaveCold *__restrict__ W)
{
    static_assert(NT == 0 || (LS && !VPK && !FULL), "the interleaved sweep table is read by the lane sweeps only");
    static_assert(!SL || !VPK, "the shell-sorted log is built for the instantiations without v-packets");
    static_assert(NT != 2 || WPE != 3, "aligned runs are eight lines long");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    WaveShared &sh = *reinterpret_cast<WaveShared *>(lds_raw);
    constexpr size_t SH_BYTES = LS ? WAVE_SHARED_LS_BYTES : sizeof(WaveShared);
    WaveSharedFull &shf = *reinterpret_cast<WaveSharedFull *>(lds_raw + SH_BYTES);
    constexpr int RING = VPK ? WV_RING_VPK : WV_RING;  // look-ahead doubles per packet
    double *ring = reinterpret_cast<double *>(lds_raw + SH_BYTES + (FULL ? sizeof(WaveSharedFull) : 0));  // [RING][64]
    double *lds_J = ring + RING * 64;
    double *lds_nubar = lds_J + H.n_shells;
    double *lds_geo = lds_nubar + H.n_shells;  // r_inner | r_outer | n_e
    int *lds_lchunk = reinterpret_cast<int *>(lds_geo + (VPK ? 4 : 3) * H.n_shells);  // SL: the open chunk of every shell (-1: none) ...
    unsigned *lds_lused = reinterpret_cast<unsigned *>(lds_lchunk + H.n_shells);       // ... and the records appended to it
    if (SL)
        for (int s = threadIdx.x; s < H.n_shells; s += 64) { lds_lchunk[s] = -1; lds_lused[s] = 0u; }
    for (int s = threadIdx.x; s < H.n_shells; s += 64) {
        lds_geo[s] = glob(W->P.r_inner)[s]; lds_geo[H.n_shells + s] = glob(W->P.r_outer)[s]; lds_geo[2 * H.n_shells + s] = glob(W->P.n_e)[s];
        if (VPK) lds_geo[3 * H.n_shells + s] = W->P.tau_rowsum ? glob(W->P.tau_rowsum)[s] : 0.0;
    }
    const int lane = threadIdx.x;  // one wave per workgroup
    {   // (volley queue: a wave keeps its partial sums from launch to launch and adds them to the estimators when it is done)
        const double *js = (W->resume && W->vq_jsave) ? W->vq_jsave + (size_t)blockIdx.x * (size_t)(2 * H.n_shells) : nullptr;
        for (int s = lane; s < 2 * H.n_shells; s += 64) lds_J[s] = js ? glob(js)[s] : 0.0;
    }

    const int j = lane & (G - 1);
    const int group_lane0 = lane & ~(G - 1);
    const double t = H.t_exp;
    const int L = H.n_lines;

    // ---- owner-lane state: this lane's packet
    Packet p;
    p.r = p.mu = p.nu = p.energy = 0.0; p.shell = 0; p.next_line_id = 0; p.status = ST_IN_PROCESS;
    int state = WS_NEED_PACKET;
    int pkt = 0;          // index of the packet in this launch's chunk
    double dop = 1.0;     // Doppler factor at the start of the prepared trace
    int pflags = 0;       // bit 0: exact-division fast path is safe; bits 1-2: delta_shell + 1
    int r_gpos = 0, r_head = 0, r_cnt = 0;  // MT19937: next state block to regenerate; LDS ring of tempered doubles
    unsigned draws = 0, events = 0, macro = 0;
    unsigned long long vvisits_total = 0;  // v-packet work counters
    unsigned vcount = 0;
    int vseq = 0;  // v-packets emitted so far by this lane's packet
    unsigned pred_bits = 0;  // roulette predictor of the volleys: bit i = v-packet i of the last volley took a roulette draw
    unsigned long long vtraced_total = 0;
    int vq_done = 0;        // volley queue / carried round: v-packets of the running volley committed so far
    bool v_out = false;     // carry-over: this lane's round of v-packets is handed over and not committed yet (its items live in this wave)
    bool v_parked = false;  // carry-over: this lane holds an unfinished item in W->vp_park
    bool vq_fresh = false;  // volley queue: this lane's round was requested in THIS launch (its results come with the next one)
    int trk_count = 0, trk_boundary = 0;  // interactions_count, boundary crossings since the last interaction
    bool trk_any = false;
    bool exhausted = false;  // wave-uniform: the chunk has no more packets to reserve
    long long res_next = 0, res_end = 0;  // wave-uniform: the block of packets this wave has reserved and not yet started
    int q_head = 0, q_tail = 0;  // wave-uniform: queue of prepared traces
    unsigned log_used = 0;  // wave-uniform: records this wave has appended to the chunk of the line-visit log it holds
    int log_chunk = -1;     // wave-uniform: that chunk (-1: none yet / the pool is empty)
    bool logged_any = false;  // wave-uniform: this launch has logged something
    unsigned long long visits = 0;
    // Profiling / test counters (reported through counters[7] under debug flags): only in the cross-check instantiations (XWALK), which the
    // host launches whenever one of those flags is set.  They are wave-uniform, i.e. SGPRs the sweep loop has none to spare of: without
    // them the headline instantiation has 53 instead of 69 spilled SGPRs, 14 instead of 25 spilled VGPRs, 204 instead of 254 lane moves.
    constexpr bool DBG = XWALK;
    unsigned dbg_rounds = 0;  // wave-uniform profiling counter: sweep rounds (reported through counters[7])
    unsigned long long dbg_vsteps = 0, dbg_vbusy = 0;  // wave-uniform profiling counters of the pooled volleys: steps of the worker loop / lanes that traced in them
    unsigned dbg_walk = 0;    // wave-uniform test counter (debug_flags 16384: jumps out of blocks longer than one window; 32768: jumps decided by the fp64 sums)
    // lane sweep (LS): the trace this lane is sweeping (it may span several passes of the event loop).  What only the exact
    // evaluation of a line needs is parked in LDS: chi in sh.d_cont0, the boundary distance in sh.d_boundary (where the
    // result goes, too); the packet's nu and Doppler factor are the owner's p.nu and dop.
    bool s_active = false, s_fast = false;
    int s_line = 0;
    unsigned s_row = 0;
    double s_tau = 0.0, s_tau_event = 0.0, s_kp = 0.0, s_xb = 0.0;
    int2 pre_blk = make_int2(0, 0);  // line_block[] of the line that stopped the trace, in flight since the sweep found it

    auto draw = [&]() {
        const double v = ring[r_head * 64 + lane];
        r_head = (r_head + 1) & (RING - 1);
        --r_cnt;
        ++draws;
        return v;
    };
    // Give every lane of `need` four more doubles: the lane regenerates the next 8 words of ITS packet's MT19937 state itself
    // (genrand's twist, mt19937.c) and parks the 4 tempered doubles in its LDS ring.  During the first pass over the state
    // the initial words it needs -- init_genrand's mt[k0 .. k0+8] and mt[k0+397 .. k0+404] -- are continued from the two
    // words kept in LDS (lazy seeding: only mt[397] was precomputed), so no memory is read at all for the first 112
    // doubles of a packet; the regenerated words go to the packet's state buffer, from which later blocks read them.
    auto refill = [&](unsigned long long need, uint32_t *seeded_states) {
        if (!((need >> lane) & 1ull)) return;
        MC_G uint32_t *st = glob(seeded_states) + ((size_t)blockIdx.x * 64 + (size_t)lane) * WV_STATE_STRIDE;  // (one state buffer per lane of the grid)
        const int k0 = r_gpos & 0x3ff;
        uint32_t wa[9], wc[8];
        // (the state words a block needs are contiguous -- mt[k0 .. k0+8] and mt[k0+397 .. k0+404] mod 624 -- except for the one
        // block whose second window straddles the end of the state: four 16-byte loads instead of seventeen 4-byte ones)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        typedef v4u v4u_a4 __attribute__((aligned(4)));
        int cb = k0 + 397;  // first word of the second window
        if (cb >= MT_N) cb -= MT_N;
        const bool c_contig = cb + 8 <= MT_N;
        auto load_c = [&]() {  // wc[0..7] = regenerated words mt[(k0 + 397 + i) mod 624]
            if (c_contig) {
                const v4u lo = *reinterpret_cast<const MC_G v4u_a4 *>(st + cb), hi = *reinterpret_cast<const MC_G v4u_a4 *>(st + cb + 4);
                wc[0] = lo.x; wc[1] = lo.y; wc[2] = lo.z; wc[3] = lo.w; wc[4] = hi.x; wc[5] = hi.y; wc[6] = hi.z; wc[7] = hi.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) wc[i] = st[(cb + i >= MT_N) ? cb + i - MT_N : cb + i];
            }
        };
        if (!(r_gpos >> 16)) {  // first pass over the state
            uint32_t w = sh.rng_a[lane];  // mt[k0]
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                wa[i] = w;
                if (i < 8) w = 1812433253u * (w ^ (w >> 30)) + (uint32_t)(k0 + i + 1);
            }
            sh.rng_a[lane] = w;  // mt[k0 + 8]
            if (k0 + 8 == MT_N) wa[8] = st[0];  // word 623 pairs with the NEW word 0
            uint32_t wb = sh.rng_b[lane];  // mt[k0 + 397]
            uint32_t wi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                wi[i] = wb;
                wb = 1812433253u * (wb ^ (wb >> 30)) + (uint32_t)(k0 + 397 + i + 1);
            }
            sh.rng_b[lane] = wb;  // mt[k0 + 405]
            if (k0 + 397 + 7 >= MT_N) {  // (part of) the second window has wrapped: those are regenerated words
                load_c();
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (k0 + 397 + i < MT_N) wc[i] = wi[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) wc[i] = wi[i];
            }
        } else {
            const v4u lo = *reinterpret_cast<const MC_G v4u *>(st + k0), hi = *reinterpret_cast<const MC_G v4u *>(st + k0 + 4);  // 32-byte aligned
            wa[0] = lo.x; wa[1] = lo.y; wa[2] = lo.z; wa[3] = lo.w; wa[4] = hi.x; wa[5] = hi.y; wa[6] = hi.z; wa[7] = hi.w;
            wa[8] = st[(k0 + 8 == MT_N) ? 0 : k0 + 8];
            load_c();
        }
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t y = (wa[i] & 0x80000000u) | (wa[i + 1] & 0x7fffffffu);
            v[i] = wc[i] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        MC_G v4u *dst = reinterpret_cast<MC_G v4u *>(st + k0);  // 32-byte aligned: k0 is a multiple of 8
        v4u lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
        dst[0] = lo4; dst[1] = hi4;
        const int tail = (r_head + r_cnt) & (RING - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[((tail + q) & (RING - 1)) * 64 + lane] = GroupRng<8>::to_double(v[2 * q], v[2 * q + 1]);
        r_cnt += 4;
        r_gpos = (k0 + 8 == MT_N) ? 0x10000 : r_gpos + 8;  // bit 16: the state has been regenerated once
    };

    if (W->resume) {  // continue where the previous epoch suspended this wave
        // (the index is made opaque so that these stay VECTOR loads: as scalar loads -- uniform address, provably global memory --
        // the wave-uniform bookkeeping below lands in SGPRs for the whole kernel and the register allocation of the event loop
        // falls apart: 45 -> 215 spilled VGPRs)
        unsigned wave_idx = blockIdx.x;
        asm volatile("" : "+v"(wave_idx));
        const WaveSave ws = gload(W->wsave + wave_idx);
        res_next = ws.res_next; res_end = ws.res_end; exhausted = ws.exhausted != 0;
        if (W->log_continue && ws.log_chunk >= 0 && ws.log_gen == W->log_gen) {  // (volley queue: many short launches share one log buffer)
            log_chunk = ws.log_chunk;
            log_used = glob(W->log.region_count)[(unsigned)log_chunk];
        }
        if (ws.done) state = WS_DONE;
        else {
            const MC_G LaneSave &v = *glob(W->save + ((size_t)blockIdx.x * 64 + lane));  // (field by field: 360 bytes at once would not fit the registers)
            p.r = v.r; p.mu = v.mu; p.nu = v.nu; p.energy = v.energy; dop = v.dop;
            s_tau = v.s_tau; s_tau_event = v.s_tau_event; s_kp = v.s_kp; s_xb = v.s_xb;
            sh.d_cont0[lane] = v.d_cont0; sh.d_boundary[lane] = v.d_boundary;
#pragma unroll
            for (int q = 0; q < RING; ++q) ring[q * 64 + lane] = v.ring[q];
            p.shell = v.shell; p.next_line_id = v.next_line; p.status = v.status; state = v.state; pkt = v.pkt; pflags = v.pflags;
            r_gpos = v.r_gpos; r_head = v.r_head; r_cnt = v.r_cnt;
            trk_count = v.trk_count; trk_boundary = v.trk_boundary;
            { const int fl = v.flags; trk_any = (fl & 1) != 0; s_active = (fl & 2) != 0; s_fast = (fl & 4) != 0; }
            s_line = v.s_line; s_row = v.s_row;
            sh.res_info[lane] = v.res_info; sh.res_line[lane] = v.res_line; pre_blk = make_int2(v.pre_blk_x, v.pre_blk_y);
            sh.rng_a[lane] = v.rng_a; sh.rng_b[lane] = v.rng_b;
            vseq = v.vseq; pred_bits = v.pred_bits; vq_done = v.vdone;
            sh.chi[lane] = v.walk_inv_new; sh.rcp_chi[lane] = v.walk_block; sh.tau_event[lane] = v.walk_event;
            sh.nu[lane] = v.trk_nu; sh.rcp_nu[lane] = v.trk_mu; sh.comov_nu[lane] = v.trk_energy;
        }
    }
    bool suspended = false, suspended_log = false, suspended_drain = false;
    unsigned dbg_passes = 0;
    // drain diagnostics (debug_flags 2097152 / 4194304 / 8388608 -> counters[7]): per wave, from the pass in which its first lane
    // found the packet supply empty: 10-ns ticks to the end of the wave / passes / live lanes summed over those passes
    unsigned long long dbg_drain_t0 = 0;
    unsigned dbg_drain_passes = 0, dbg_drain_lanes = 0;
#ifdef TMC_SECTION_TIMERS  // profiling builds only: wall time of the sections of a pass, section (debug_flags >> 8) & 7 -> counters[7]
    unsigned long long sec_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long sec_prev = __builtin_amdgcn_s_memtime();
#define TMC_SEC(i) { if (H.debug_flags & 4096) __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sec_t[i] += now_ - sec_prev; sec_prev = now_; }
#else
#define TMC_SEC(i)
#endif
    for (;;) {
        // ============================================================ event phase (lane-per-packet)
        if (DBG) ++dbg_passes;
        asm volatile("" ::: "memory");  // the cold arguments are (re)loaded here, once per pass
        const GroupArgs &P = W->P;
        const EstimatorLog &log = W->log;
        uint32_t *const seeded_states = W->seeded_states;
        const long long chunk_first = W->chunk_first, chunk_count = W->chunk_count;
        double *const jb = P.jblue_t, *const ed = P.edot_t;
        // a pass appends at most 64 records: without room for them the wave takes the next chunk of the pool (one atomic per chunk)
        bool log_full = false;
        // SL: lane s looks after shell s's open chunk: one without room for a whole pass (64 records, should every lane log into that shell) is closed and
        // replaced from the pool -- so a pass never runs out of room half-way, and a wave that finds the pool empty suspends before anything of the pass has
        // happened.  (Ranking this pass's records per shell up front instead -- ballots over the shell's bits, exact reservations, chunks opened on demand --
        // was built first and measured: +3 % instructions in every pass, propagation +4 %, which took back all the estimator passes gained;
        // profiles/r06_shell_sorted_log.txt.)
        if (SL && log.region_capacity > 0 && __ballot(state != WS_DONE) != 0ull) {
            bool need = false;
            if (lane < H.n_shells) need = lds_lchunk[lane] < 0 || lds_lused[lane] + 64u > log.region_capacity;
            if (__ballot(need)) {  // (once per chunk and shell)
                bool full = false;
                if (need) {
                    if (lds_lchunk[lane] >= 0) glob(log.region_count)[lds_lchunk[lane]] = min(lds_lused[lane], log.region_capacity);
                    const unsigned c = gatomic_add_u32(log.pool_next, 1u);
                    full = c >= (unsigned)log.n_regions;
                    lds_lchunk[lane] = full ? -1 : (int)c;
                    lds_lused[lane] = 0u;
                }
                log_full = __ballot(full) != 0ull && W->save != nullptr;  // the pool is empty: the epoch is over for this wave
            }
        }
        if (!SL && log.region_capacity > 0 && (log_chunk < 0 || log_used + 64 > log.region_capacity) && __ballot(state != WS_DONE) != 0ull) {
            if (lane == 0 && log_chunk >= 0) glob(log.region_count)[log_chunk] = min(log_used, log.region_capacity);
            unsigned c = 0;
            if (lane == 0) c = gatomic_add_u32(log.pool_next, 1u);
            c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
            log_used = 0;
            if (c < (unsigned)log.n_regions) log_chunk = (int)c;
            else { log_chunk = -1; log_full = W->save != nullptr; }  // the pool is empty: the epoch is over for this wave
        }
        bool vq_stop = false;
        if (VPK && W->vq_items) {
            // volley queue: lanes whose round is with the tracer cannot go on in this launch; once (nearly) all others have
            // joined them the wave suspends -- the tracer runs between this launch and the next
            const unsigned long long waiting = __ballot(state == WS_VOLLEY && vq_fresh);
            const unsigned long long can = __ballot(state != WS_DONE && !(state == WS_VOLLEY && vq_fresh));
            vq_stop = waiting != 0ull && __popcll(can) <= H.vq_min_active;
        }
        // (drain_split = T: once T or fewer lanes are left -- 64: as soon as the supply has run out)
        const int drain_live = W->drain_split ? __popcll(__ballot(state != WS_DONE)) : 0;
        const bool drain_stop = W->drain_split && W->save && exhausted && res_next == res_end && logged_any && drain_live != 0 && drain_live <= W->drain_split;
        if (log_full || vq_stop || drain_stop) {
            // this wave's region of the line-visit log is full (or its lanes wait for the v-packet tracer): suspend the lanes as
            // they are (every lane is at the top of a pass: a swept trace waiting for its event, a lane sweep in progress, a
            // requested volley round, or done) and leave the rest to the next epoch
            suspended_log = log_full;
            suspended_drain = drain_stop && !log_full && !vq_stop;
            MC_G LaneSave &v = *glob(W->save + ((size_t)blockIdx.x * 64 + lane));
            v.r = p.r; v.mu = p.mu; v.nu = p.nu; v.energy = p.energy; v.dop = dop;
            v.s_tau = s_tau; v.s_tau_event = s_tau_event; v.s_kp = s_kp; v.s_xb = s_xb;
            v.d_cont0 = sh.d_cont0[lane]; v.d_boundary = sh.d_boundary[lane];
#pragma unroll
            for (int q = 0; q < WV_RING_VPK; ++q) v.ring[q] = q < RING ? ring[q * 64 + lane] : 0.0;
            v.shell = p.shell; v.next_line = p.next_line_id; v.status = p.status; v.state = state; v.pkt = pkt; v.pflags = pflags;
            v.r_gpos = r_gpos; v.r_head = r_head; v.r_cnt = r_cnt;
            v.trk_count = trk_count; v.trk_boundary = trk_boundary;
            v.flags = (trk_any ? 1 : 0) | (s_active ? 2 : 0) | (s_fast ? 4 : 0);
            v.s_line = s_line; v.s_row = s_row;
            v.res_info = sh.res_info[lane]; v.res_line = sh.res_line[lane]; v.pre_blk_x = pre_blk.x; v.pre_blk_y = pre_blk.y;
            v.rng_a = sh.rng_a[lane]; v.rng_b = sh.rng_b[lane];
            v.vseq = vseq; v.pred_bits = pred_bits; v.vdone = vq_done; v.pad_v0 = v.pad_v1 = v.pad_v2 = 0;
            v.walk_inv_new = sh.chi[lane]; v.walk_block = sh.rcp_chi[lane]; v.walk_event = sh.tau_event[lane]; v.pad_w = 0.0;
            v.trk_nu = sh.nu[lane]; v.trk_mu = sh.rcp_nu[lane]; v.trk_energy = sh.comov_nu[lane];
            suspended = true;
            break;
        }
        // every live packet gets the draws of one pass: new direction, first macro-atom jump, next tau_event
        const bool ready = state == WS_SWEEP && !(LS && s_active);  // the prepared trace has been swept
        // (up to the capacity of the ring, so that the refills inside the walk and before the prologue -- each a dependent round
        // trip of its own -- are rarely needed)
        refill(__ballot((ready || state == WS_NEED_TRACE || state == WS_WALK) && r_cnt <= RING - 4), seeded_states);
        int err = 0, type = 0, emit = -1, mb0 = 0, mb1 = 0;
        double inv_new = 1.0, distance = 0.0, emit_nu_walk = 0.0;
        bool in_macro = false, interacted = false;
        bool want_volley = false;  // this lane's packet launches a volley of v-packets in this pass
        // a macro-atom walk that the previous pass left unfinished (the wave does not wait for its longest chains): the line
        // interaction it belongs to is complete up to the jump, what the rest of the pass needs was parked in LDS
        const bool resumed = state == WS_WALK;
        if (resumed) {
            in_macro = true; interacted = true; type = IT_LINE;
            inv_new = sh.chi[lane];
            const int2 blk = reinterpret_cast<const int2 *>(sh.rcp_chi)[lane];
            mb0 = blk.x; mb1 = blk.y;
        }
        // ---- log the line visits of the finished traces (update_line_estimators, deferred: estimator_log.hpp)
        {
            int n_visit = 0, start = 0;
            if (ready) {
                const int info = sh.res_info[lane];
                distance = sh.d_boundary[lane];
                const int code = info & 7;
                start = p.next_line_id;
                const int stop_line = (info & 8) ? L : sh.res_line[lane];
                n_visit = stop_line - start + ((code == 3) ? 1 : 0);
                // (lane sweeps: the lines a trace examined -- those it passed and, unless the list ran out, the one that stopped it -- are counted here, once
                // per trace, instead of step by step inside the sweep loop)
                if (LS) visits += (unsigned long long)(unsigned)(stop_line - start + ((info & 8) ? 0 : 1));
                if (!(info & 8)) p.next_line_id = stop_line;
                if (code == 4) err = ERR_MONTECARLO;
                type = code == 1 ? IT_BOUNDARY : (code == 2 ? IT_ESCATTERING : IT_LINE);
                if (DBG && (P.debug_flags & 1)) n_visit = 0;
            }
            // every wave appends to the chunk it holds: no atomics, no empty slots
            const unsigned long long have = __ballot(n_visit > 0);
            if (have) {
                unsigned my = 0;
                int my_chunk = log_chunk;
                if (SL) {  // the next slot of the shell's open chunk (an LDS atomic hands the lanes of a pass that share a shell distinct slots)
                    logged_any = true;
                    if (n_visit > 0) { my = atomicAdd(&lds_lused[p.shell], 1u); my_chunk = lds_lchunk[p.shell]; }
                } else {
                    my = log_used + (unsigned)__popcll(have & ((1ull << lane) - 1ull));
                    log_used += (unsigned)__popcll(have);
                    logged_any = true;
                }
                if (n_visit > 0) {
                    LineVisitRecord rec;
                    const double inv_nu = 1.0 / p.nu;
                    rec.c_e = FULL ? p.energy : p.energy * inv_nu;
                    rec.c_jb = rec.c_e * inv_nu;
                    rec.idx0 = (unsigned)p.shell * (unsigned)L + (unsigned)start;
                    rec.n = (unsigned)n_visit;
                    if (my_chunk >= 0 && my < log.region_capacity) {
                        const size_t slot = (size_t)(unsigned)my_chunk * log.region_capacity + my;
                        gstore(log.records + slot, rec);
                        glob(log.keys)[slot] = (unsigned)(p.shell * log.tiles_per_shell + start / EST_TILE);
                    } else {  // no log (or no chunk to be had and no way to suspend): add the terms directly (slow path)
                        for (int k = 0; k < n_visit; ++k) {
                            const double f = FULL ? 1.0 : glob(P.nu_line)[(unsigned)(start + k)];
                            gatomic_add_f64(&jb[rec.idx0 + (unsigned)k], rec.c_jb * f);
                            gatomic_add_f64(&ed[rec.idx0 + (unsigned)k], rec.c_e * f);
                        }
                    }
                }
            }
        }
        TMC_SEC(0)
        // ---- epilogue of the finished traces: move, estimators, boundary / scattering
        if (ready && !err) {
            // move_r_packet + update_estimators_bulk (packets/movement.py:31-76)
            const double r = p.r;
            if (distance > 0.0) {
                const double new_r = sqrt(r * r + distance * distance + 2.0 * r * distance * p.mu);
                const double mu_new = (p.mu * r + distance) / new_r;
                const double comov_nu = p.nu * dop;
                const double comov_energy = p.energy * dop;
                const double dist_est = FULL ? distance * dop : distance;
                if (!(DBG && (P.debug_flags & 2))) {
                    atomicAdd(&lds_J[p.shell], comov_energy * dist_est);
                    atomicAdd(&lds_nubar[p.shell], comov_energy * dist_est * comov_nu);
                }
                p.mu = mu_new;
                p.r = new_r;
            }
            if (type == IT_BOUNDARY) {
                if (TRACK) trk_boundary += 1;
                cross_shell(p.shell, p.status, ((pflags >> 1) & 3) - 1, P.n_shells);
            } else {
                interacted = true;
                if (TRACK) {
                    // the packet before the interaction, parked in LDS (free between the sweeps) until the record is
                    // written once the interaction is complete: nothing of it occupies registers during the macro-atom walk
                    sh.nu[lane] = p.nu; sh.rcp_nu[lane] = p.mu; sh.comov_nu[lane] = p.energy;
                }
                // common part of line_scatter_event (interaction_event_callers.py:187-239) and thomson_scatter
                // (interaction_events.py:184-217): Doppler with the old angle, new isotropic angle, Doppler back
                const double vel = p.r / t;
                const double old_dop = doppler_factor<FULL>(vel, p.mu);
                const double scat_comov_nu = p.nu * old_dop;
                const double comov_energy = p.energy * old_dop;
                p.mu = 2.0 * draw() - 1.0;
                inv_new = inverse_doppler_factor<FULL>(vel, p.mu);
                p.energy = comov_energy * inv_new;
                if (type != IT_LINE) p.nu = scat_comov_nu * inv_new;
                if (type == IT_LINE) {
                    emit = p.next_line_id;
                    if (P.line_interaction_type != 0) {
                        const int2 blk = LS ? pre_blk : gload(P.line_block + (unsigned)p.next_line_id);
                        mb0 = blk.x; mb1 = blk.y;
                        in_macro = true;
                    }
                }
            }
        }
        TMC_SEC(1)
        // ---- macro_atom_interaction (macro_atom.py:52-104): one jump of every walking packet per round.  The reference adds
        // the block's probabilities up until the sum exceeds the drawn number; the sums are precomputed (cum_t, same
        // additions in the same order), so the jump is the first entry of the block's monotone run that exceeds it:
        // the first eight entries in one round trip (the short blocks of downbranch end there), then a 4-ary search.
        if (P.cum16) {
            // macroatom mode, the default: every lane walks for its own packet on the compact tables (walk_tables.hpp) -- per
            // jump one 16..64-byte read of the block's 16-bit running sums (all of a block of <= 32 transitions; longer blocks
            // are first narrowed by a binary search over their 8-entry quads) and one 8-byte read of what the selected
            // transition leads to.  64 jumps of a wave are in flight at once; here mb0 / mb1 = compact start / rows of the block.
            // A block with a hot sector (walk_tables.hpp; mb1 < 0, mb0 = block id) is entered through that sector: ONE request decides
            // the jump and names its destination when the number drawn falls into one of the block's six widest intervals; else
            // the same number is looked up in the block's own tables in the next round (`redo`).
            const int walk_cut = (DBG && (H.debug_flags & 16777216)) ? H.walk_min_active : (H.walk_min_active * __popcll(__ballot(state != WS_DONE))) >> 6;
            bool redo = false;
            double event = 0.0;
            if (resumed && mb1 >= 0 && (mb1 & WALK_REDO)) { redo = true; mb1 &= ~WALK_REDO; event = sh.tau_event[lane]; }
            for (int round = 0;; ++round) {
                const unsigned long long walking = __ballot(in_macro);
                if (!walking) break;
                // the wave does not wait for its longest chains (a geometric tail: each jump ends a walk with the probability
                // of an emission): once few lanes are still walking and others wait (their interaction is complete, or their
                // sweep goes on), those few carry their walk over to the next pass
                // (with v-packets only in the group-sweep instantiations: there the pooled volleys keep their item list in the sweep queue's
                // LDS, idle between the sweep phases; in the lane-sweep ones it lies over the arrays in which a carried walk parks its interaction)
                // (the cut-off is a share of the wave's LIVE lanes: in the drain of a call -- a wave with a handful of packets left --
                // a fixed count would end the phase after every round and charge each of them a whole pass)
                if ((!VPK || !LS) && round > 0 && __popcll(walking) <= walk_cut && __ballot(!in_macro && state != WS_DONE)) break;
                refill(__ballot(in_macro && !redo && r_cnt < 1), seeded_states);
                unsigned x = 0;
                if (in_macro) {
                    if (!redo) event = draw();
                    x = (unsigned)(event * 65536.0);  // floor: event is in [0, 1)
                }
                const bool hot = in_macro && mb1 < 0, cold = in_macro && mb1 >= 0;
                int q_lo = 0, q_hi = cold ? (mb1 + 7) >> 3 : 0;
                const MC_G unsigned short *__restrict__ blk = glob(P.cum16) + ((size_t)p.shell * P.cum16_stride + (unsigned)(cold ? mb0 : 0));
                if (DBG && (H.debug_flags & 16384)) dbg_walk += (unsigned)__popcll(__ballot(cold && mb1 > 8 * WALK_WINDOW_QUADS));  // tests: jumps out of long blocks
                // blocks of more than 32 transitions: first quad whose last entry is not below x (entries are monotone)
                while (__ballot(cold && mb1 > 8 * WALK_WINDOW_QUADS && q_hi - q_lo > WALK_WINDOW_QUADS - 1)) {
                    if (cold && mb1 > 8 * WALK_WINDOW_QUADS && q_hi - q_lo > WALK_WINDOW_QUADS - 1) {
                        const int qm = q_lo + ((q_hi - q_lo) >> 1);
                        const unsigned v = blk[8 * qm + 7];
                        if (v < x) q_lo = qm + 1; else q_hi = qm;
                    }
                }
                // ---- first round trip of the jump: the window of a cold block / the sector of a hot one (same registers)
                int sel = -1;  // position of the selected transition in its block; -1: the block ran out
                bool exact = false;
                int nxt = 0;   // second round trip: 1 rec16[mb0 + sel], 2 blk_tab[nxt_arg], 3 nu_line[emit]
                unsigned nxt_arg = 0;
                if (in_macro) {
                    const int nq = hot ? 4 : min(WALK_WINDOW_QUADS, ((mb1 + 7) >> 3) - q_lo);
                    typedef unsigned u4v __attribute__((ext_vector_type(4)));
                    const MC_G u4v *__restrict__ wp = hot ? reinterpret_cast<const MC_G u4v *>(glob(P.hot_sec) + ((size_t)p.shell * (size_t)P.hot_stride + (size_t)(unsigned)mb0 * 16))
                                                          : reinterpret_cast<const MC_G u4v *>(blk + 8 * q_lo);
                    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
                    if (nq > 0) { const u4v t = wp[0]; w0 = make_uint4(t.x, t.y, t.z, t.w); }
                    if (nq > 1) { const u4v t = wp[1]; w1 = make_uint4(t.x, t.y, t.z, t.w); }
                    if (nq > 2) { const u4v t = wp[2]; w2 = make_uint4(t.x, t.y, t.z, t.w); }
                    if (nq > 3) { const u4v t = wp[3]; w3 = make_uint4(t.x, t.y, t.z, t.w); }
                    if (hot) {
                        const unsigned lo2[3] = {w0.x, w0.y, w0.z}, hi2[3] = {w0.w, w1.x, w1.y}, k2[3] = {w3.x, w3.y, w3.z};
                        const unsigned dw[HOT_ENTRIES] = {w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
                        bool hit = false;
                        unsigned d = 0, k16 = 0;
#pragma unroll
                        for (int e = 0; e < HOT_ENTRIES; ++e) {
                            const unsigned sh16 = 16u * (unsigned)(e & 1);
                            const unsigned l = (lo2[e >> 1] >> sh16) & 0xffffu, h = (hi2[e >> 1] >> sh16) & 0xffffu;
                            if (x >= l && x < h) { hit = true; d = dw[e]; k16 = (k2[e >> 1] >> sh16) & 0xffffu; }  // (the intervals are disjoint)
                        }
                        if (hit) {
                            macro += k16 + 1u;
                            if (d & WALK_EMIT) { emit = (int)(d & 0x7fffffffu); in_macro = false; nxt = 3; }
                            else if (d & WALK_HOT_DEST) mb0 = (int)(d & 0x3fffffffu);
                            else { nxt = 2; nxt_arg = d; }
                        } else { redo = true; nxt = 2; nxt_arg = (unsigned)mb0; }
                    } else {
                        const unsigned xx = x | (x << 16);
                        unsigned less = 0, gt = 0;
                        if (nq > 0) walk_count_quad(w0, xx, less, gt);
                        if (nq > 1) walk_count_quad(w1, xx, less, gt);
                        if (nq > 2) walk_count_quad(w2, xx, less, gt);
                        if (nq > 3) walk_count_quad(w3, xx, less, gt);
                        const int n_less = (int)((less & 0xffffu) + (less >> 16)), n_gt = (int)((gt & 0xffffu) + (gt >> 16));
                        int k = 8 * q_lo + n_less;  // every entry before it is surely <= the number drawn
                        if (8 * nq - n_gt - n_less > 0) {
                            exact = true;
                            // entries equal to x (2^-16 of the draws per entry; the 0xffff padding when x = 65535): the reference's
                            // own comparison on the fp64 running sums, in order
                            const MC_G double *__restrict__ cum = glob(P.cum_t) + (size_t)p.shell * (size_t)P.n_trans;
                            for (; k < mb1; ++k) {
                                const unsigned v = blk[k];
                                if (v > x) break;
                                const int2 qi = gload(P.quad_info + ((unsigned)(mb0 + k) >> 3));
                                if (cum[(unsigned)(qi.x + (k & 7))] > event) break;
                            }
                        }
                        if (k < mb1) sel = k;
                        macro += (unsigned)(sel >= 0 ? sel + 1 : mb1);
                        redo = false;
                        if (sel < 0) { err = ERR_MACRO_ATOM; in_macro = false; }
                        else nxt = 1;
                    }
                }
                if (DBG && (H.debug_flags & 32768)) dbg_walk += (unsigned)__popcll(__ballot(exact));  // tests: jumps decided by the fp64 sums
                if (DBG && (H.debug_flags & 65536)) dbg_walk += (unsigned)__popcll(__ballot(hot && !redo));  // tests: jumps decided by a hot sector
                if (DBG && (H.debug_flags & 131072)) dbg_walk += (unsigned)__popcll(__ballot(hot && redo));  // tests: numbers a hot sector did not decide
                // ---- second round trip: what the selected transition leads to
                if (nxt == 1) {
                    const WalkRec rec = gload(P.rec16 + (unsigned)(mb0 + sel));
                    if (rec.b & WALK_EMIT) {
                        emit = (int)rec.a;
                        emit_nu_walk = rec.nu;  // (the emission line's frequency travels with the record: no read of nu_line afterwards)
                        in_macro = false;
                        if (rec.b & WALK_UNSUPPORTED) err = ERR_UNSUPPORTED;
                    } else { mb0 = (int)rec.a; mb1 = (rec.b & WALK_HOT) ? -1 : (int)rec.b; }
                } else if (nxt == 2) {
                    const int2 bt = gload(P.blk_tab + nxt_arg);
                    mb0 = bt.x; mb1 = bt.y;
                } else if (nxt == 3)
                    emit_nu_walk = (NT != 0 && emit < L - 1) ? glob(P.nt_t)[2u * ((unsigned)p.shell * P.nt_stride + (unsigned)emit)]
                                                             : glob(P.nu_line)[(unsigned)emit];  // (the sector the coming sweep starts in; the interleaved table
                                                                                                  // holds -inf in the frequency slot of the last line)
            }
            if (in_macro) {  // carried over
                state = WS_WALK;
                sh.chi[lane] = inv_new;
                reinterpret_cast<int2 *>(sh.rcp_chi)[lane] = make_int2(mb0, redo ? (mb1 | WALK_REDO) : mb1);
                if (redo) sh.tau_event[lane] = event;
            }
        } else if (XWALK && P.line_interaction_type == 2 && !(P.debug_flags & 128)) {  // (flag 128: the per-lane search below, for tests)
            // macroatom mode (long chains of jumps, and a wave waits for its longest chain: one coalesced round trip per jump): the wave's G-lane groups scan the blocks, G
            // probabilities per coalesced load, accumulated in the reference's serial order (macro_atom_group() of the
            // group kernel).  The work items and results live in LDS that the trace parameters do not need right now.
            double *mac_event = sh.nu;
            int *mac_row = reinterpret_cast<int *>(sh.rcp_nu), *mac_b0 = mac_row + 64;
            int *mac_b1 = reinterpret_cast<int *>(sh.comov_nu), *mac_q = mac_b1 + 64;
            int *res_emit = reinterpret_cast<int *>(sh.chi), *res_type = res_emit + 64;
            int *res_b0 = reinterpret_cast<int *>(sh.rcp_chi), *res_b1 = res_b0 + 64;
            int *res_cnt = reinterpret_cast<int *>(sh.tau_event);
            const int gshift = lane & ~(G - 1);
            constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : ((G == 8) ? 0xffull : 0xfull);
            for (;;) {
                const unsigned long long items = __ballot(in_macro);
                if (!items) break;
                refill(__ballot(in_macro && r_cnt < 1), seeded_states);
                const int n_items = __popcll(items);
                if (in_macro) {
                    mac_q[__popcll(items & ((1ull << lane) - 1ull))] = lane;
                    mac_event[lane] = draw();
                    mac_row[lane] = p.shell * P.n_trans;
                    mac_b0[lane] = mb0; mac_b1[lane] = mb1;
                }
                // rounds of 64/G items; the first chunk of the next round's item (probabilities and transition records) is
                // loaded before the current one is scanned
                auto item_of = [&](int k, int &o, unsigned &row, int &b0, int &b1, double &event) {
                    o = mac_q[k]; event = mac_event[o]; row = (unsigned)mac_row[o]; b0 = mac_b0[o]; b1 = mac_b1[o];
                };
                int n_o = -1, n_b0 = 0, n_b1 = 0;
                unsigned n_row = 0;
                double n_event = 0.0, n_pr = 0.0;
                int4 n_rec = make_int4(0, 0, 0, 0);
                if (lane / G < n_items) {
                    item_of(lane / G, n_o, n_row, n_b0, n_b1, n_event);
                    const int kk = n_b0 + j;
                    if (kk < n_b1) { n_pr = glob(P.cum_t)[n_row + (unsigned)kk]; n_rec = gload(P.trans_rec + (unsigned)kk); }
                }
                for (int base = 0; base < n_items; base += 64 / G) {
                    const int o = n_o, b0 = n_b0, b1 = n_b1;
                    const unsigned row = n_row;
                    const double event = n_event;
                    double pr = n_pr;
                    int4 rec = n_rec;
                    const bool have = base + lane / G < n_items;
                    // prefetch the next round's item
                    n_o = -1; n_pr = 0.0; n_rec = make_int4(0, 0, 0, 0);
                    if (base + 64 / G + lane / G < n_items) {
                        item_of(base + 64 / G + lane / G, n_o, n_row, n_b0, n_b1, n_event);
                        const int kk = n_b0 + j;
                        if (kk < n_b1) { n_pr = glob(P.cum_t)[n_row + (unsigned)kk]; n_rec = gload(P.trans_rec + (unsigned)kk); }
                    }
                    if (have) {
                        int cnt = 0;
                        bool found = false;
                        int4 hit_rec = make_int4(0, 0, 0, 0);
                        for (int b = b0; b < b1; b += G) {
                            const int kk = b + j;
                            const bool in = kk < b1;
                            if (b != b0) {
                                pr = in ? glob(P.cum_t)[row + (unsigned)kk] : 0.0;
                                rec = in ? gload(P.trans_rec + (unsigned)kk) : make_int4(0, 0, 0, 0);
                            }
                            const double acc = pr;  // the block's running sum up to this entry (cum_t): nothing to scan
                            const unsigned hit = (unsigned)((__ballot(in && acc > event) >> gshift) & GMASK);
                            if (hit) {
                                const int f = __builtin_ctz(hit);
                                cnt += f + 1;
                                hit_rec.x = gbcast<G>(rec.x, f); hit_rec.y = gbcast<G>(rec.y, f);
                                hit_rec.z = gbcast<G>(rec.z, f); hit_rec.w = gbcast<G>(rec.w, f);
                                found = true;
                                break;
                            }
                            const int n_in = min(G, b1 - b);
                            cnt += n_in;
                        }
                        if (j == 0) {
                            if (found) {
                                res_emit[o] = hit_rec.x; res_type[o] = hit_rec.y; res_b0[o] = hit_rec.z; res_b1[o] = hit_rec.w;
                                res_cnt[o] = cnt;
                            } else res_cnt[o] = -cnt - 1;
                        }
                    }
                }
                if (in_macro) {
                    const int c = res_cnt[lane];
                    if (c < 0) { macro += (unsigned)(-c - 1); err = ERR_MACRO_ATOM; in_macro = false; }
                    else {
                        macro += (unsigned)c;
                        emit = res_emit[lane]; mb0 = res_b0[lane]; mb1 = res_b1[lane];
                        const int tt = res_type[lane];
                        if (tt < 0) {
                            in_macro = false;
                            if (tt != -1) err = ERR_UNSUPPORTED;
                        }
                    }
                }
            }
        }
        const bool carried = state == WS_WALK && in_macro;  // (set above, compact walk only)
        if (carried) in_macro = false;
        // downbranch (one jump over a short block)
        bool have_emit_nu = P.cum16 != nullptr && emit >= 0 && !carried;  // (the compact walk brought the frequency along)
        double emit_nu = emit_nu_walk;
        if (XWALK) {
            bool searching = false;  // the first eight entries did not decide this lane's jump: 4-ary search in [lo, hi)
            int lo = 0, hi = 0;      // cum[j] <= event for all block entries j < lo; hi == mb1 or cum[hi] > event
            double event = 0.0;
            while (__ballot(in_macro)) {
                refill(__ballot(in_macro && !searching && r_cnt < 1), seeded_states);
                if (in_macro) {
                    const MC_G double *__restrict__ cum = glob(P.cum_t) + (unsigned)p.shell * (unsigned)P.n_trans;
                    int k = -2;  // >= 0: the jump goes to transition k; -1: no entry exceeds the number drawn; -2: search on
                    if (!searching) {
                        // the first eight running sums of the block in one round trip (the table carries eight entries of
                        // slack, entries past the end of the block are ignored)
                        event = draw();
                        const MC_G double *__restrict__ c8 = cum + (unsigned)mb0;
                        double a8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) a8[q] = c8[q];
                        k = -1;
#pragma unroll
                        for (int q = 7; q >= 0; --q)
                            if (mb0 + q < mb1 && a8[q] > event) k = mb0 + q;
                        if (k < 0 && mb1 - mb0 > 8) { searching = true; lo = mb0 + 8; hi = mb1; k = -2; }
                    } else {
                        const int n = hi - lo;
                        const int i0 = lo + (n >> 2), i1 = lo + (n >> 1), i2 = lo + ((3 * n) >> 2);
                        const double a0 = cum[(unsigned)i0], a1 = cum[(unsigned)i1], a2 = cum[(unsigned)i2];
                        if (a0 > event) hi = i0;
                        else if (a1 > event) { lo = i0 + 1; hi = i1; }
                        else if (a2 > event) { lo = i1 + 1; hi = i2; }
                        else lo = i2 + 1;
                        if (lo >= hi) { searching = false; k = lo < mb1 ? lo : -1; }
                    }
                    if (k == -1) { macro += (unsigned)(mb1 - mb0); err = ERR_MACRO_ATOM; in_macro = false; }
                    else if (k >= 0) {
                        macro += (unsigned)(k - mb0 + 1);
                        const int4 rec = gload(P.trans_rec + (unsigned)k);
                        emit_nu = glob(P.trans_nu)[(unsigned)k]; have_emit_nu = true;
                        emit = rec.x; mb0 = rec.z; mb1 = rec.w;
                        if (rec.y < 0) {
                            in_macro = false;
                            if (rec.y != -1) err = ERR_UNSUPPORTED;
                        }
                    }
                }
            }
        }
        TMC_SEC(2)
        // ---- finish the interaction, hand finished packets over
        if ((ready || resumed) && !carried) {
            if (interacted && !err) {
                int emit_id = -1;
                const int absorb_id = (type == IT_LINE) ? p.next_line_id : -1;  // (the line that absorbed the packet)
                if (type == IT_LINE) {  // line_emission (interaction_events.py:227-258); its inverse Doppler factor == inv_new
                    p.nu = (have_emit_nu ? emit_nu : glob(P.nu_line)[emit]) * inv_new;
                    p.next_line_id = emit + 1;
                    emit_id = emit;
                }
                if (FULL) p.mu = aberration_cmf_to_lf(p.r, t, p.mu);
                if (TRACK) {
                    trk_count += 1 + trk_boundary;
                    trk_boundary = 0;
                    trk_any = true;
                    TrackerRecord rec;
                    rec.before_nu = sh.nu[lane]; rec.before_mu = sh.rcp_nu[lane]; rec.before_energy = sh.comov_nu[lane];
                    rec.radius = p.r; rec.after_mu = p.mu;
                    rec.shell = p.shell; rec.type = type; rec.absorb = absorb_id; rec.emit = emit_id;
                    rec.count = trk_count; rec.valid = 1;
                    gstore(reinterpret_cast<TrackerRecord *>(W->D.li_rec) + (chunk_first + pkt), rec);
                }
            }
            state = WS_NEED_TRACE;
            // volley after a line or electron-scattering interaction (classic/packet_propagation.py:201-244)
            if (VPK && interacted && !err) want_volley = true;
            if (err || p.status != ST_IN_PROCESS) {
                const long long i = chunk_first + pkt;
                const DeviceProblem *C = &W->D;
                if (err) {
                    gatomic_min_i64(&C->first_error[0], i);
                    glob(C->out_nu)[i] = (double)err;
                    glob(C->out_e)[i] = -99.0;
                } else {
                    // set_packet_collection_output (modes/montecarlo_transport.py:70-90)
                    glob(C->out_nu)[i] = p.nu;
                    glob(C->out_e)[i] = (p.status == ST_REABSORBED) ? -p.energy : p.energy;
                    if (TRACK && !trk_any) {  // no interaction at all: an empty record (the unpacking fills in NaN / -1 / 0)
                        TrackerRecord rec;
                        rec.before_nu = rec.before_mu = rec.before_energy = rec.radius = rec.after_mu = 0.0;
                        rec.shell = rec.type = rec.absorb = rec.emit = -1; rec.count = 0; rec.valid = 0;
                        gstore(reinterpret_cast<TrackerRecord *>(C->li_rec) + i, rec);
                    }
                }
                state = WS_NEED_PACKET;
            }
        }
        TMC_SEC(3)
        // ---- fetch packets (one global atomic per wave and pass)
        {
            const unsigned long long need_pkt = __ballot(state == WS_NEED_PACKET);
            if (need_pkt) {
                // Packets are handed out from a block the wave has reserved (one atomic per WV_RESERVE packets, not per pass):
                // the first lanes take what is left of the current block, the others start the next one.
                const int n_want = __popcll(need_pkt);
                const int n_old = (int)min((long long)n_want, res_end - res_next);
                const long long old_next = res_next;
                res_next += n_old;
                long long new_base = chunk_count;
                const int n_new = n_want - n_old;
                if (n_new > 0 && !exhausted) {
                    const int n_res = max(WV_RESERVE, n_new);
                    unsigned long long b = 0;
                    if (lane == 0) b = gatomic_add_u64(P.next_packet, (unsigned long long)n_res);
                    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b);
                    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
                    const long long got = (long long)(((unsigned long long)bhi << 32) | blo);
                    if (got + n_res >= chunk_count) exhausted = true;
                    new_base = min(got, chunk_count);
                    res_next = min(new_base + n_new, chunk_count);
                    res_end = min(new_base + n_res, chunk_count);
                }
                if (state == WS_NEED_PACKET) {
                    const int rank = __popcll(need_pkt & ((1ull << lane) - 1ull));
                    const long long mine = rank < n_old ? old_next + rank : new_base + (rank - n_old);
                    if (mine >= chunk_count) state = WS_DONE;
                    else {
                        pkt = (int)mine;
                        r_gpos = r_head = r_cnt = 0;
                        const MC_G LaunchRec &lr = *glob(W->launch + mine);
                        sh.rng_a[lane] = lr.seed;        // mt[0]
                        sh.rng_b[lane] = lr.checkpoint;  // mt[397]
                        p.r = lr.r; p.mu = lr.mu; p.nu = lr.nu; p.energy = lr.energy;
                        p.next_line_id = lr.line0;
                        p.shell = 0; p.status = ST_IN_PROCESS;
                        if (TRACK) { trk_count = 0; trk_boundary = 0; trk_any = false; }  // (-1 + the initial track_boundary_event)
                        state = WS_NEED_TRACE;
                        // volley at launch (classic/packet_propagation.py:109-118).  The roulette predictor of a new packet: where the
                        // screening is on (long line lists on fine grids: nearly every v-packet is dropped by the roulette) its first volley
                        // is predicted dropped, too -- and therefore screened, and its draw positions come out right at the first attempt:
                        // -3 % on the configs[4] shape (profiles/r05_vpacket_requests.txt).  Scheduling only: a wrong prediction is re-traced.
                        // (debug flag 2048: the predictor of rounds 1-4, for A/B)
                        if (VPK) { vseq = 0; pred_bits = (P.tau_pfx && !(P.debug_flags & 2048)) ? 0xffffffffu : 0u; want_volley = true; }
                    }
                }
            }
        }
        if (DBG && (H.debug_flags & (2097152 | 4194304 | 8388608))) {
            const unsigned long long live = __ballot(state != WS_DONE);
            if (!dbg_drain_t0 && __popcll(live) < 64) dbg_drain_t0 = wall_clock64();
            if (dbg_drain_t0) { ++dbg_drain_passes; dbg_drain_lanes += (unsigned)__popcll(live); }
        }
        if (__ballot(state != WS_DONE) == 0ull) break;
        if (VPK) {
            // ---- trace_vpacket_volley (virtual_packet.py:248-386): all lanes with a volley trace their i-th v-packet together
            const int n_v = (int)P.n_vpackets;
            int verr = 0;
            const DeviceProblem *C = &W->D;
            // the owner validates and commits the v-packets of a round in order (both volley paths)
            auto commit_round = [&](const VpResult *res, const int n_round, int &vdone) {
                bool valid = true;
                int n_ok = 0, consumed = 0;
                unsigned obs = 0;
                for (int sl = 0; sl < n_round; ++sl) {
                    const VpResult r = gload(res + sl);
                    vtraced_total += (unsigned)r.visits;
                    if (r.used > 0) obs |= 1u << sl;  // learn from every trace of the round, committed or not
                    if (valid) {
                        if (r.err) { verr = r.err; valid = false; }
                        else {
                            ++vcount;
                            vvisits_total += (unsigned)r.visits;
                            // add_vpacket_collection_to_histogram (modes/montecarlo_transport.py:166-195)
                            if (!(r.nu < P.grid0 || r.nu > P.grid_last) && r.energy != 0.0) {  // (a dropped v-packet adds 0.0: the bin keeps its bits)
                                const long long idx = (long long)floor((r.nu - P.grid0) / P.delta_nu);
                                gatomic_add_f64(&P.vhist[idx], r.energy);
                            }
                            if (C->vlog_count) {
                                const unsigned long long slot = gatomic_add_u64(C->vlog_count, 1ull);
                                if ((long long)slot < C->vlog_capacity) {
                                    glob(C->vlog_packet)[slot] = chunk_first + pkt; glob(C->vlog_seq)[slot] = vseq;
                                    glob(C->vlog_nu)[slot] = r.nu; glob(C->vlog_energy)[slot] = r.energy; glob(C->vlog_mu)[slot] = r.mu0; glob(C->vlog_r)[slot] = p.r;
                                }
                            }
                            ++vseq;
                            ++n_ok;
                            consumed += 1 + r.used;
                            // it started from the right stream position itself; the items after it did only if it
                            // consumed what was predicted
                            if (r.used != (int)((pred_bits >> (vdone + sl)) & 1u)) valid = false;
                        }
                    }
                }
                const unsigned seen = (1u << n_round) - 1u;
                pred_bits = (pred_bits & ~(seen << vdone)) | (obs << vdone);
                r_head = (r_head + consumed) & (RING - 1);
                r_cnt -= consumed;
                draws += (unsigned)consumed;
                vdone += n_ok;
            };
            // volley queue: the round this lane requested in the previous launch has come back from the tracer
            bool cont_volley = false;  // ... and the queue has been switched off since (the drain of a call): the rest of the volley is pooled
            if (state == WS_VOLLEY && !vq_fresh) {
                commit_round(W->vp_scratch + ((size_t)blockIdx.x * 64 + lane) * VP_ROUND, min(VP_ROUND, n_v - vq_done), vq_done);
                if (verr || vq_done == n_v) state = WS_NEED_TRACE;  // the volley is complete: on to the next trace
                else if (!W->vq_items) { state = WS_NEED_TRACE; cont_volley = true; }
            }
            // carry-over: the round this lane handed over in an earlier pass is still with the wave's workers -- or, after a suspension
            // (the workers' items are not saved), has to be handed over again
            const bool v_carried = state == WS_VCARRY && v_out;
            if (state == WS_VCARRY && !v_out) { state = WS_NEED_TRACE; cont_volley = true; }
            bool in_volley = cont_volley || v_carried || (want_volley && state == WS_NEED_TRACE && !(p.nu < P.spawn_start || p.nu > P.spawn_end) && n_v > 0);
            double mu_min = 0.0, beta_inner = 0.0, mu_bin = 0.0, r_dop = 1.0;
            bool on_inner = false;
            if (in_volley) {
                const double r_in0 = lds_geo[0];
                if (p.r > r_in0) {
                    const double r_inner_over_r = r_in0 / p.r;
                    mu_min = -sqrt(1 - r_inner_over_r * r_inner_over_r);
                    if (FULL) mu_min = aberration_lf_to_cmf(p.r, t, mu_min);
                } else {
                    on_inner = true;
                    if (FULL) {
                        const double inv_c = 1 / C_LIGHT;
                        const double inv_t = 1 / t;
                        beta_inner = r_in0 * inv_t * inv_c;
                    }
                }
                mu_bin = (1.0 - mu_min) / (double)n_v;
                r_dop = doppler_factor<FULL>(p.r / t, p.mu);
            }
            // Pooled volley.  The v-packets of ALL volleys of the wave are work items for ALL 64 lanes, so a packet deep in
            // the ejecta (many shells per v-packet) does not make the lanes of shallow packets wait.  The n_v mu-draws and
            // the Russian-roulette draws come from the parent's stream in sequence, so an item reads its draws at the
            // position predicted by the packet's bit mask of "v-packet i played roulette last time" (same mu bin, same
            // optical depth to first order -- the group kernel's predictor); the owner lane then commits its items in
            // order up to and including the first one whose draw consumption differs from the prediction, and the rest is
            // traced again in the next round from the corrected stream position.
            if (W->vq_items) {
                // ---- volley queue: commit the round that came back from the tracer, request the next one (see VolleyRequest)
                const size_t slot = (size_t)blockIdx.x * 64 + lane;
                if (in_volley) vq_done = 0;  // a new volley
                const bool ask = !verr && (in_volley || (state == WS_VOLLEY && !vq_fresh));  // (or the next round of a running one)
                const int n_round = ask ? min(VP_ROUND, n_v - vq_done) : 0;
                for (;;) {  // one mu draw and one roulette draw per v-packet of the round must be in the ring
                    const unsigned long long need = __ballot(ask && r_cnt < 2 * n_round);
                    if (!need) break;
                    refill(need, seeded_states);
                }
                int incl = n_round;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off);
                    if (lane >= off) incl += up;
                }
                const int n_items = __shfl(incl, 63);
                if (n_items > 0) {
                    unsigned base = 0;
                    if (lane == 0) base = gatomic_add_u32(W->vq_count, (unsigned)n_items);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (ask) {
                        VolleyRequest rq;
                        rq.r = p.r; rq.mu = p.mu; rq.nu = p.nu; rq.energy = p.energy;
                        rq.shell = p.shell; rq.next_line = p.next_line_id; rq.vdone = vq_done; rq.cnt = r_cnt;
                        rq.pred_bits = pred_bits; rq.pad0 = rq.pad1 = rq.pad2 = 0;
#pragma unroll
                        for (int q = 0; q < WV_RING_VPK; ++q) rq.draws[q] = ring[((r_head + q) & (RING - 1)) * 64 + lane];
                        gstore(W->vq_req + slot, rq);
                        MC_G unsigned *it = glob(W->vq_items) + (base + (unsigned)(incl - n_round));
                        for (int sl = 0; sl < n_round; ++sl) it[sl] = ((unsigned)slot << 3) | (unsigned)sl;
                        state = WS_VOLLEY;
                        vq_fresh = true;
                    }
                }
            } else {
            VpResult *vres = W->vp_scratch + (size_t)blockIdx.x * (64 * VP_ROUND);  // slot: owner lane * VP_ROUND + v-packet of the round
            // [64 * VP_ROUND] work items: over sh.cursor | sh.rowfast | sh.queue (group sweeps: every prepared trace has been swept, the
            // queue is empty) or over sh.nu | sh.rcp_nu (lane sweeps: no walk is carried there, so nothing is parked in them now)
            unsigned short *items = LS ? reinterpret_cast<unsigned short *>(sh.nu) : reinterpret_cast<unsigned short *>(sh.cursor);
            int vdone = (cont_volley || v_carried) ? vq_done : 0;  // v-packets of this lane's volley committed so far
            // Cut-off with carry-over (W->vp_carry_min_active > 0): a phase does not wait for its longest v-packets.  Once no item
            // waits, at most that many lanes still trace and some packet of the wave could go on, the phase ends; the tracing lanes
            // park their v-packets (VpPark) and take them up again in the next pass's phase, their owners wait in WS_VCARRY with the
            // round uncommitted (nothing of it is committed before all of it is there: the order of the reference).
            const int carry_cut = W->vp_park ? W->vp_carry_min_active : 0;
            // ---- worker state (lives across the rounds of a phase; across passes in W->vp_park)
            bool tracing = false;
            int w_owner = 0, w_item = 0, w_q = 0, w_used = 0, w_avail = 0, w_head = 0;
            unsigned my_visits = 0;
            VpState vs;
            double v_rcp_nu = 0.0;
            bool v_fast = false;
            vs.r = vs.mu = vs.nu = vs.energy = vs.tau = vs.mu0 = 0.0; vs.shell = 0; vs.next_line = 0;
            // screening (tau_prefix.hpp): an item predicted to be dropped by the roulette is first traced on the prefix sums; if
            // that does not decide it, it starts again line by line from its launch state (v0_*)
            bool screening = false;
            double v_margin = 0.0, v0_r = 0.0, v0_energy = 0.0;
            int v0_shell = 0, v0_line = 0;
            if (v_parked) {
                const VpPark k = gload(W->vp_park + ((size_t)blockIdx.x * 64 + lane));
                vs.r = k.r; vs.mu = k.mu; vs.nu = k.nu; vs.energy = k.energy; vs.tau = k.tau; vs.mu0 = k.mu0; vs.shell = k.shell; vs.next_line = k.next_line;
                v_rcp_nu = k.rcp_nu; v_margin = k.margin; v0_r = k.v0_r; v0_energy = k.v0_energy; v0_shell = k.v0_shell; v0_line = k.v0_line;
                w_owner = k.owner; w_item = k.item; w_q = k.q; w_used = k.used; w_avail = k.avail; w_head = k.head;
                my_visits = k.visits; v_fast = (k.flags & 1) != 0; screening = (k.flags & 2) != 0;
                tracing = true;
                v_parked = false;
            }
            int phase_steps = 0;
            bool cut = false;  // the worker loop was left with lanes still tracing
            while (__ballot(in_volley)) {
                const bool publish = in_volley && !v_out;
                const int n_round = publish ? min(VP_ROUND, n_v - vdone) : 0;
                for (;;) {  // one mu draw and one roulette draw per v-packet of the round must be in the ring
                    const unsigned long long need = __ballot(publish && r_cnt < 2 * n_round);
                    if (!need) break;
                    refill(need, seeded_states);
                }
                // work items of the round: (owner lane << 8) | slot
                int incl = n_round;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off);
                    if (lane >= off) incl += up;
                }
                const int item0 = incl - n_round;
                const int n_items = __shfl(incl, 63);
                for (int sl = 0; sl < n_round; ++sl) items[item0 + sl] = (unsigned short)((lane << 8) | sl);
                if (publish) v_out = true;
                // ---- workers
                int next = 0;
                cut = false;
                for (;;) {
                    const unsigned long long free_l = __ballot(!tracing);
                    if (carry_cut > 0 && next >= n_items && phase_steps >= 4) {
                        // (only when somebody gains by it: a packet that needs no volley, or whose volley is complete, can go on)
                        const int busy = 64 - __popcll(free_l);
                        if (busy > 0 && busy <= carry_cut && __ballot(state != WS_DONE && !in_volley) != 0ull) { cut = true; break; }
                    }
                    const int n_take = min(__popcll(free_l), n_items - next);
                    const int rank = __popcll(free_l & ((1ull << lane) - 1ull));
                    const bool take = !tracing && rank < n_take;
                    const int my_item = next + rank;
                    next += n_take;
                    if (!__ballot(tracing || take)) break;
                    ++phase_steps;
                    if (DBG && (H.debug_flags & (134217728 | 268435456))) { ++dbg_vsteps; dbg_vbusy += (unsigned long long)__popcll(__ballot(tracing || take)); }
                    if (n_take > 0) {  // wave-uniform: some lane starts an item and needs its owner's state
                        const int it = take ? (int)items[my_item] : 0;
                        const int o = take ? (it >> 8) : lane;
                        const int slot = it & 0xff;
                        const double f_r = __shfl(p.r, o), f_nu = __shfl(p.nu, o), f_energy = __shfl(p.energy, o), f_rdop = __shfl(r_dop, o);
                        const double f_mu_min = __shfl(mu_min, o), f_mu_bin = __shfl(mu_bin, o), f_beta = __shfl(beta_inner, o);
                        const int f_shell = __shfl(p.shell, o), f_line = __shfl(p.next_line_id, o), f_inner = __shfl((int)on_inner, o);
                        const int f_vdone = __shfl(vdone, o), f_head = __shfl(r_head, o), f_cnt = __shfl(r_cnt, o);
                        const unsigned f_pred = (unsigned)__shfl((int)pred_bits, o);
                        if (take) {
                            const int i = f_vdone + slot;  // index of the v-packet in its volley
                            const unsigned before = (f_pred >> f_vdone) & ((1u << slot) - 1u);  // predicted roulette draws of slots < slot
                            const int q = slot + __popc(before);
                            const double xi = ring[((f_head + q) & (RING - 1)) * 64 + o];
                            w_owner = o; w_item = o * VP_ROUND + slot; w_q = q + 1; w_used = 0; w_avail = f_cnt; w_head = f_head;
                            double v_mu = f_mu_min + (double)i * f_mu_bin + xi * f_mu_bin;
                            double weight;
                            if (f_inner) {
                                if (!FULL) weight = 2 * v_mu / (double)n_v;
                                else weight = 2 * (v_mu + f_beta) / (2 * f_beta + 1) / (double)n_v;
                            } else
                                weight = (1 - f_mu_min) / (double)(2 * n_v);
                            if (FULL) v_mu = aberration_cmf_to_lf(f_r, t, v_mu);
                            const double v_dop = doppler_factor<FULL>(f_r / t, v_mu);
                            const double ratio = f_rdop / v_dop;
                            vs.r = f_r; vs.mu = v_mu; vs.mu0 = v_mu;  // the log records the (aberrated) launch direction (:337-340,375)
                            vs.nu = f_nu * ratio;
                            v_rcp_nu = 1.0 / vs.nu;
                            v_fast = mid_range(vs.nu);
                            vs.energy = f_energy * weight * ratio;
                            vs.tau = 0.0; vs.shell = f_shell; vs.next_line = f_line;
                            my_visits = 0;
                            tracing = true;
                            screening = P.tau_pfx != nullptr && ((f_pred >> i) & 1u) != 0u;
                            v_margin = 0.0; v0_r = f_r; v0_energy = vs.energy; v0_shell = f_shell; v0_line = f_line;
                        }
                    }
                    if (tracing) {
                        auto wdraw = [&]() {
                            const double d = ring[((w_head + w_q + w_used) & (RING - 1)) * 64 + w_owner];
                            ++w_used;
                            return d;
                        };
                        int draws_left = w_avail - (w_q + w_used);
                        int st;
                        if (screening) {
                            st = vp_screen_step<FULL>(P, wdraw, draws_left, vs, v_margin, v_rcp_nu, v_fast, lds_geo, my_visits);
                            if (st == 1 && (P.debug_flags & 67108864)) vtraced_total += 1ull << 40;  // tests: decided on the prefix sums -> counters[7] >> 40
                            if (st == 2) {  // not decided on the prefix sums: again, line by line
                                vs.r = v0_r; vs.mu = vs.mu0; vs.energy = v0_energy; vs.tau = 0.0; vs.shell = v0_shell; vs.next_line = v0_line;
                                my_visits = 0; w_used = 0; screening = false; st = 0;
                            }
                        } else
                            st = vp_shell_step<FULL>(P, wdraw, draws_left, vs, v_rcp_nu, v_fast, lds_geo, my_visits);
                        if (st != 0) {
                            VpResult r;
                            r.nu = vs.nu; r.energy = st == 1 ? vs.energy * mcm::exp(-vs.tau) : 0.0; r.mu0 = vs.mu0;
                            r.used = w_used; r.visits = (int)my_visits; r.err = st < 0 ? st : 0; r.pad = 0;
                            gstore(vres + w_item, r);
                            tracing = false;
                        }
                    }
                }
                // ---- the owners whose round is complete validate and commit their items in order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int pending = 0;  // items of this lane's round that a worker still traces
                for (unsigned long long m = __ballot(tracing); m; m &= m - 1) {
                    const int q = __builtin_ctzll(m);
                    if (__builtin_amdgcn_readlane(w_owner, q) == lane) ++pending;
                }
                if (in_volley && v_out && pending == 0) {
                    commit_round(vres + lane * VP_ROUND, min(VP_ROUND, n_v - vdone), vdone);
                    v_out = false;
                    if (state == WS_VCARRY) state = WS_NEED_TRACE;  // (a carried round: the packet goes on once its volley is complete)
                    if (verr || vdone == n_v) in_volley = false;
                }
                // (left by the cut-off: on to the next pass unless an owner can hand over another round right away)
                if (cut && !__ballot(in_volley && !v_out)) break;
            }
            if (in_volley && v_out) {  // this lane's round stays with the workers: the packet waits for it
                state = WS_VCARRY;
                vq_done = vdone;
            }
            if (tracing) {
                VpPark k;
                k.r = vs.r; k.mu = vs.mu; k.nu = vs.nu; k.energy = vs.energy; k.tau = vs.tau; k.mu0 = vs.mu0; k.shell = vs.shell; k.next_line = vs.next_line;
                k.rcp_nu = v_rcp_nu; k.margin = v_margin; k.v0_r = v0_r; k.v0_energy = v0_energy; k.v0_shell = v0_shell; k.v0_line = v0_line;
                k.owner = w_owner; k.item = w_item; k.q = w_q; k.used = w_used; k.avail = w_avail; k.head = w_head;
                k.visits = my_visits; k.flags = (v_fast ? 1 : 0) | (screening ? 2 : 0);
                gstore(W->vp_park + ((size_t)blockIdx.x * 64 + lane), k);
                v_parked = true;
            }
            }
            if (verr) {  // the reference raises: the packet ends with the error code
                const long long i = chunk_first + pkt;
                gatomic_min_i64(&C->first_error[0], i);
                glob(C->out_nu)[i] = (double)verr;
                glob(C->out_e)[i] = -99.0;
                state = WS_NEED_PACKET;
            }
        }
        TMC_SEC(4)
        // ---- prologue of the next trace (trace_packet, modes/homologous_rad_packet_transport.py:30-98)
        refill(__ballot(state == WS_NEED_TRACE && r_cnt < 1), seeded_states);
        {
            const bool go = state == WS_NEED_TRACE;
            if (go) {
                const double velocity = p.r / t;
                dop = doppler_factor<FULL>(velocity, p.mu);
                double chi_e = lds_geo[2 * P.n_shells + p.shell] * P.sigma_thomson;
                if (FULL) chi_e *= dop;
                double d_boundary;
                int delta;
                distance_boundary(p.r, p.mu, lds_geo[p.shell], lds_geo[P.n_shells + p.shell], d_boundary, delta);
                const double tau_event = -mcm::log(draw());
                const double comov_nu = p.nu * dop;
                ++events;
                const bool fast = mid_range(p.nu) && mid_range(chi_e) && mid_range(tau_event) && mid_range(p.energy) &&
                                  mid_range(p.r) && mid_range(comov_nu) && mid_range(P.t_exp) && !(DBG && (P.debug_flags & 4));
                pflags = (fast ? 1 : 0) | ((delta + 1) << 1);
                if (LS) {
                    sh.d_cont0[lane] = chi_e; sh.d_boundary[lane] = d_boundary;
                    s_tau_event = tau_event;
                    s_tau = 0.0; s_line = p.next_line_id; s_row = (unsigned)p.shell * (NT != 0 ? P.nt_stride : (unsigned)L);
                    s_kp = ((chi_e * P.tc) / p.nu) * (1.0 + 0x1p-40);
                    s_xb = ((d_boundary * p.nu) * P.rcp_tc) * (1.0 - 0x1p-40);
                    s_fast = fast && mid_range(s_kp);
                    s_active = true;
                } else {
                    sh.nu[lane] = p.nu; sh.rcp_nu[lane] = 1.0 / p.nu; sh.comov_nu[lane] = comov_nu;
                    sh.chi[lane] = chi_e; sh.rcp_chi[lane] = 1.0 / chi_e;
                    sh.tau_event[lane] = tau_event; sh.d_boundary[lane] = d_boundary;
                    sh.d_cont0[lane] = tau_event / chi_e;  // distance_continuum in force at the first line
                    if (FULL) { shf.r[lane] = p.r; shf.mu[lane] = p.mu; }
                    sh.cursor[lane] = p.next_line_id;
                    sh.rowfast[lane] = (int)(((unsigned)p.shell * (unsigned)L) | (fast ? 0x80000000u : 0u));
                }
                state = WS_SWEEP;
            }
            if (!LS) {
                const unsigned long long go_mask = __ballot(go);
                if (go) sh.queue[(q_tail + __popcll(go_mask & ((1ull << lane) - 1ull))) & 63] = lane;
                q_tail += __popcll(go_mask);
            }
        }

        TMC_SEC(5)
        // ============================================================ sweep phase, lane sweep: every lane its own trace
        if (LS) {
            // (like the walk's: the cut-off of the sweep phase is a share of the live lanes, see there)
            const int ls_cut = (DBG && (H.debug_flags & 16777216)) ? H.ls_min_active : (H.ls_min_active * __popcll(__ballot(state != WS_DONE))) >> 6;
            for (int step = 0;; ++step) {
                const unsigned long long act = __ballot(s_active);
                if (!act) break;
                if (step > 0) {
                    // Finished lanes wait for the event phase, which costs the same however few lanes take part in it: go
                    // on sweeping until enough of the wave has something to do there.
                    const unsigned long long waiting = __ballot(state == WS_SWEEP && !s_active);
                    if (waiting && (__popcll(act) <= ls_cut || step >= H.ls_max_steps)) break;
                }
                if (DBG) ++dbg_rounds;
                if (s_active) {
                    const double *__restrict__ pn = H.nu_line + (unsigned)s_line;
                    const double *__restrict__ pt = H.tau_t + (s_row + (unsigned)s_line);
                    // (experiment: twelve lines per step in the 150-VGPR instantiation, option ls_waves_per_simd = 3; the tables carry 16 lines of slack)
                    constexpr int CH = (WPE == 3 && !VPK) ? 12 : ((NT == 1 && WPE == 4) ? LS_CHUNK_NT : LS_CHUNK);
                    double nl[CH], tl[CH];
                    typedef double nt2 __attribute__((ext_vector_type(2)));
                    const unsigned nt_at = s_row + (unsigned)s_line;
                    const nt2 *__restrict__ pq = reinterpret_cast<const nt2 *>(H.tau_t) + (NT == 2 ? (nt_at & ~7u) : nt_at);
                    const int k0 = NT == 2 ? (int)(nt_at & 7u) : 0;  // entries of the aligned run in front of the current line
                    if constexpr (NT != 0) {
#pragma unroll
                        for (int k = 0; k < CH; ++k) { const nt2 e = pq[k]; nl[k] = e.x; tl[k] = e.y; }
                    } else {
#pragma unroll
                        for (int k = 0; k < CH; ++k) { nl[k] = pn[k]; tl[k] = pt[k]; }
                    }
                    // (experiment, cross-check instantiations only, debug flag 524288: touch the last optical depth of the NEXT chunk with this
                    // chunk's loads, so that the sibling sector of a 128-byte line is requested together with the first and the next step's
                    // loads find their sectors on the way -- VERDICT r04 "next" 4; profiles/r05_tau_sibling_touch.txt)
                    if (DBG && (H.debug_flags & 524288)) {
                        const unsigned touch = reinterpret_cast<const unsigned *>(pt + min(2 * LS_CHUNK - 1, L - 1 + LS_CHUNK - s_line))[1];
                        if (touch == 0x7ff00001u) ++dbg_rounds;  // (never: an optical depth is not that NaN pattern; keeps the load alive)
                    }
                    int adv = 0;  // lines of this chunk the trace has passed
                    const double comov = p.nu * dop;
                    // lines that provably do not stop the trace (see above); the first one that might is kept in f_nu / f_tau
                    const int n_fast = L - 1 - s_line;  // lines of the chunk before the last line of the list
                    bool alive;
                    double f_nu, f_tau;
                    if constexpr ((WPE == 3 && !VPK) || (TMC_STRAIGHT_A != 0 && WPE == 4 && NT == 1)) {
                        // Straight-line form (the twelve-line instantiation only): the reference's serial sums of the chunk first (t[k] = optical
                        // depth in front of line k), then the four bounds of every line -- independent of each other, no exec masking, no branch
                        // per line -- into one bit per line; the first set bit is the line the loop below would have stopped at (same operands,
                        // same operations: same decision), and the lines after it cost a few flops nobody reads.  Measured
                        // (profiles/r05_straight_line_sweep.txt): the sweep loop shrinks from 601 to 448 instructions (SALU 268 -> 134, a branch per
                        // line -> none); calls that are mostly drain -3.5 % (1.25e7 packets of configs[2]: 572 -> 552 ms), but a full chip +2 ... +4 %
                        // in BOTH instantiations -- the branch per line skips the rest of a chunk once every lane still sweeping has stopped,
                        // which late in a sweep phase (a handful of lanes left) is most steps -- so the sixteen-wave instantiation keeps the loop.
                        double t[CH + 1];
                        t[0] = s_tau;
#pragma unroll
                        for (int k = 0; k < CH; ++k) t[k + 1] = t[k] + tl[k];
                        unsigned fail = 0;
#pragma unroll
                        for (int k = CH - 1; k >= 0; --k) {
                            const double X = comov - nl[k];
                            const double x = s_kp * X;
                            const double sum = t[k + 1] + x;
                            bool ok;
                            if constexpr (NT != 0) ok = (k > 0 || X >= 0.0) && X < s_xb && sum < s_tau_event;  // the lean proof (see the loop form below)
                            else {
                                const double D = s_tau_event - t[k];
                                ok = X >= 0.0 && X < s_xb && x < D && sum <= s_tau_event;
                            }
                            fail = fail + fail + (ok ? 0u : 1u);
                        }
                        // (the last line of the list and everything behind it, and every line of a trace outside mid_range, go to the exact evaluation; on the
                        // interleaved table the end of the list fails by itself: -inf in its frequency slot)
                        if constexpr (NT != 0) fail |= s_fast ? (1u << CH) : 1u;
                        else fail |= 1u << (s_fast ? min(max(n_fast, 0), CH) : 0);
                        adv = __builtin_ctz(fail);
                        alive = adv == CH;
                        double ts = t[0];
#pragma unroll
                        for (int k = 1; k <= CH; ++k)
                            if (adv >= k) ts = t[k];
                        s_tau = ts;
                        // (the stopping line's frequency and optical depth: read again -- the sectors were fetched a moment ago -- instead of
                        // selected from registers that would have to stay alive for it)
                        if constexpr (NT != 0) { const nt2 e = pq[alive ? 0 : adv]; f_nu = e.x; f_tau = e.y; }
                        else { f_nu = pn[alive ? 0 : adv]; f_tau = pt[alive ? 0 : adv]; }
                    } else {
                        alive = s_fast;
                        f_nu = nl[0]; f_tau = tl[0];
                        if constexpr (NT == 2) {  // (a trace outside mid_range evaluates its current line exactly: entry k0 of the run)
#pragma unroll
                            for (int k = 1; k < CH; ++k)
                                if (k == k0) { f_nu = nl[k]; f_tau = tl[k]; }
                        }
                        int kf = alive ? CH : k0;  // first entry of the run that did not pass (the lines passed: kf - k0)
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            if (alive && (NT != 2 || k >= k0)) {
                                const double X = comov - nl[k];
                                const double x = s_kp * X;
                                const double tau_n = s_tau + tl[k];
                                const double sum = tau_n + x;
                                bool ok;
                                if constexpr (NT != 0) {
                                    // The lean form of the no-stop proof (round 6).  On the interleaved table (built by the host only for tables without a negative
                                    // optical depth, with -inf in the frequency slot of the last line of the list and of everything behind it) three of the five
                                    // tests of the general form below are implied by the other two:
                                    //   * k < n_fast: the last line and the slack fail X < X_b (X = +inf) and go to the exact evaluation, which does not look at the
                                    //     frequency of the last line;
                                    //   * X >= 0 (the reference's "nu difference" error) for k > 0: the list is sorted, X grows along it (first entry of a run: tested);
                                    //   * x < RN(tau_event - tau_prev) (no electron-scattering stop): with tau_line >= 0, RN(tau_prev + x) <= RN(RN(tau_prev + tau_line) + x)
                                    //     = sum < tau_event gives tau_prev + x < tau_event exactly, hence x <= RN(tau_event - tau_prev), and x exceeds RN(chi d_trace) by
                                    //     2^-40 relative (K'): d_trace < RN(RN(tau_event - tau_prev) / chi) = d_continuum still follows.  (sum < tau_event, strict:
                                    //     with <= the first step would only give tau_prev + x <= tau_event + half an ulp.)
                                    // 22 instead of 31 instructions per line; tests/test_lane_sweep_bounds.py checks the implication on 2e6 inputs within ulps of every threshold.
                                    ok = (k > (NT == 2 ? k0 : 0) || X >= 0.0) && X < s_xb && sum < s_tau_event;
                                } else {
                                    const double D = s_tau_event - s_tau;
                                    ok = k - k0 < n_fast && X >= 0.0 && X < s_xb && x < D && sum <= s_tau_event;
                                }
                                if (ok) s_tau = tau_n;
                                else { alive = false; kf = k; f_nu = nl[k]; f_tau = tl[k]; }
                            }
                        }
                        adv = kf - k0;
                    }
                    if (!alive) {
                        // that line with the reference's own arithmetic (or the for-else, once the list is exhausted)
                        const double chi = sh.d_cont0[lane], d_bound = sh.d_boundary[lane];
                        const int line = s_line + adv;
                        int code;
                        double dist;
                        if (line < L) {
                            code = lane_exact_line(H, line, f_nu, f_tau, s_tau, p.nu, comov, chi, s_tau_event, d_bound, dist);
                            if (!code) { s_tau = s_tau + f_tau; ++adv; }
                        } else {
                            // for-else (lines 157-172): the line list is exhausted; next_line_id is left untouched (bit 3)
                            const double d_cont = (s_tau_event - s_tau) / chi;
                            const bool cont = d_cont < d_bound;
                            dist = cont ? d_cont : d_bound;
                            code = (cont ? 2 : 1) | 8;
                        }
                        if (code) {
                            sh.d_boundary[lane] = dist; sh.res_info[lane] = code; sh.res_line[lane] = (code & 8) ? 0 : line;
                            if (code == 3 && H.line_block) pre_blk = H.line_block[(unsigned)line];
                            s_active = false;
                        }
                    }
                    s_line += adv;
                }
            }
            TMC_SEC(6)
            continue;
        }

        // ============================================================ sweep phase (G-lane groups work off the queue)
        // worker state of this lane's group; every group is idle again when the phase ends
        SweepSlot cur;
        cur.owner = -1; cur.cur0 = 0; cur.row = 0; cur.fast = true;
        cur.nu = cur.rcp_nu = cur.comov_nu = cur.chi = cur.rcp_chi = cur.tau_event = cur.d_boundary = cur.r = cur.mu = 0.0;
        cur.tau_carry = cur.d_cont_carry = cur.nu_line = cur.tau_line = 0.0;
        int nxt_owner = -1, n_cursor = 0;
        unsigned n_rowfast = 0;
        double n_nu = 0.0, n_tau = 0.0;
        // The phase ends when every prepared trace has been swept.  (Ending it earlier, as soon as most lanes have their
        // result, and letting the long sweeps run on across event phases was measured: fewer but fuller steps, more event
        // phases, no gain.)
        for (;;) {
            const int avail = q_tail - q_head;
            const unsigned long long busy = __ballot(cur.owner >= 0 || nxt_owner >= 0);
            if (avail == 0 && !busy) break;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): everything prefetched during the previous step has landed
            if (DBG) ++dbg_rounds;
            // ---- a group whose sweep ended continues with the trace whose first lines it prefetched
            if (cur.owner < 0 && nxt_owner >= 0) {
                const int o = nxt_owner;
                cur.owner = o;
                nxt_owner = -1;
                cur.nu = sh.nu[o]; cur.rcp_nu = sh.rcp_nu[o]; cur.comov_nu = sh.comov_nu[o];
                cur.chi = sh.chi[o]; cur.rcp_chi = sh.rcp_chi[o]; cur.tau_event = sh.tau_event[o]; cur.d_boundary = sh.d_boundary[o];
                if (FULL) { cur.r = shf.r[o]; cur.mu = shf.mu[o]; }
                cur.fast = (n_rowfast >> 31) != 0;
                cur.cur0 = n_cursor; cur.row = n_rowfast & 0x7fffffffu;
                cur.nu_line = n_nu; cur.tau_line = n_tau;
                cur.tau_carry = 0.0;
                cur.d_cont_carry = sh.d_cont0[o];
            }
            // ---- hand waiting traces to the groups without a prefetched one and start loading their first lines
            {
                const unsigned long long free_groups = __ballot(nxt_owner < 0 && j == 0);
                if (avail > 0 && free_groups) {
                    const int rank = __popcll(free_groups & ((1ull << group_lane0) - 1ull));
                    if (nxt_owner < 0 && rank < avail) {
                        nxt_owner = sh.queue[(q_head + rank) & 63];
                        n_cursor = sh.cursor[nxt_owner];
                        n_rowfast = (unsigned)sh.rowfast[nxt_owner];
                        const int line = n_cursor + j;
                        const bool in = line < L;
                        n_nu = in ? H.nu_line[(unsigned)line] : 0.0;
                        n_tau = in ? H.tau_t[(n_rowfast & 0x7fffffffu) + (unsigned)line] : 0.0;
                    }
                    q_head += min(__popcll(free_groups), avail);
                }
            }
            // ---- one G-line step of every running sweep
            if (cur.owner >= 0) {
                if (cur.fast) sweep_step<FULL, G, true>(H, cur, j, sh, visits);
                else sweep_step<FULL, G, false>(H, cur, j, sh, visits);
            }
        }
        TMC_SEC(6)
    }

    if (!SL && lane == 0 && W->log.region_capacity > 0 && log_chunk >= 0) glob(W->log.region_count)[log_chunk] = min(log_used, W->log.region_capacity);
    if (SL && W->log.region_capacity > 0)
        for (int s = lane; s < H.n_shells; s += 64)
            if (lds_lchunk[s] >= 0) glob(W->log.region_count)[lds_lchunk[s]] = min(lds_lused[s], W->log.region_capacity);
    const DeviceProblem *C = &W->D;
    const bool keep = suspended && W->vq_jsave != nullptr;  // volley queue: partial sums and counters stay with the wave
    if (W->vq_jsave) {
        double *js = W->vq_jsave + (size_t)blockIdx.x * (size_t)(2 * H.n_shells);
        for (int s = lane; s < 2 * H.n_shells; s += 64) glob(js)[s] = keep ? lds_J[s] : 0.0;
    }
    if (!keep)
        for (int s = lane; s < H.n_shells; s += 64) {
            if (lds_J[s] != 0.0) gatomic_add_f64(&C->J[s], lds_J[s]);
            if (lds_nubar[s] != 0.0) gatomic_add_f64(&C->nubar[s], lds_nubar[s]);
        }
    // counters: wave-reduce, one atomic each
    unsigned long long v = (LS || j == 0) ? visits : 0ull;  // group sweeps: group-uniform, count once per group
    unsigned long long e = events, m = macro, d = draws, vv = vvisits_total, vc = vcount;
    unsigned long long vt = (H.debug_flags & (134217728 | 268435456)) ? 0ull : vtraced_total;  // (those flags report something else through counters[7])
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off); e += __shfl_down(e, off); m += __shfl_down(m, off); d += __shfl_down(d, off);
        vv += __shfl_down(vv, off); vc += __shfl_down(vc, off); vt += __shfl_down(vt, off);
    }
    if (lane == 0) {
        unsigned long long cn[7] = {v, e, m, d, vv, vc, vt};
        if (W->wsave) {
            WaveSave ws;
            ws.res_next = res_next; ws.res_end = res_end; ws.exhausted = exhausted ? 1 : 0; ws.done = suspended ? 0 : 1;
            ws.log_chunk = log_chunk; ws.log_gen = W->log_gen;
            if (W->vq_jsave && W->resume) {
                const WaveSave old = gload(W->wsave + blockIdx.x);
#pragma unroll
                for (int k = 0; k < 7; ++k) cn[k] += old.cnt[k];
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) ws.cnt[k] = keep ? cn[k] : 0ull;
            gstore(W->wsave + blockIdx.x, ws);
            if (suspended) gatomic_add_u32(W->suspended, 1u);
            if (suspended_log) gatomic_add_u32(W->suspended + 1, 1u);
            if (suspended_drain) gatomic_add_u32(W->suspended + 2, 1u);
        }
        if (!keep) {
            if (cn[0]) gatomic_add_u64(&C->counters[0], cn[0]);
            if (cn[1]) gatomic_add_u64(&C->counters[1], cn[1]);
            if (cn[2]) gatomic_add_u64(&C->counters[2], cn[2]);
            if (cn[3]) gatomic_add_u64(&C->counters[5], cn[3]);
            if (VPK) {
                if (cn[4]) gatomic_add_u64(&C->counters[3], cn[4]);
                if (cn[5]) gatomic_add_u64(&C->counters[4], cn[5]);
                if (cn[6]) gatomic_add_u64(&C->counters[7], cn[6]);
            }
        }
        if (H.debug_flags & 16) gatomic_add_u64(&C->counters[7], (unsigned long long)dbg_rounds);  // profiling only
        if (VPK && (H.debug_flags & 134217728)) gatomic_add_u64(&C->counters[7], dbg_vbusy);    // profiling only: lane-steps of the pooled volleys' workers
        if (VPK && (H.debug_flags & 268435456)) gatomic_add_u64(&C->counters[7], dbg_vsteps);   // profiling only: steps of the pooled volleys' worker loop
        if (H.debug_flags & (16384 | 32768 | 65536 | 131072)) gatomic_add_u64(&C->counters[7], (unsigned long long)dbg_walk);  // tests only
        if (H.debug_flags & 2097152) gatomic_add_u64(&C->counters[7], dbg_drain_t0 ? wall_clock64() - dbg_drain_t0 : 0ull);
        if (H.debug_flags & 4194304) gatomic_add_u64(&C->counters[7], (unsigned long long)dbg_drain_passes);
        if (H.debug_flags & 8388608) gatomic_add_u64(&C->counters[7], (unsigned long long)dbg_drain_lanes);
        if (H.debug_flags & 32) gatomic_add_u64(&C->counters[7], (unsigned long long)dbg_passes);
#ifdef TMC_SECTION_TIMERS
        if (H.debug_flags & 64) {
            unsigned long long tsel = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (((H.debug_flags >> 8) & 7) == i) tsel = sec_t[i];
            gatomic_add_u64(&C->counters[7], tsel);
        }
#endif
    }
}

// ---- volley queue: the v-packet tracer (see VolleyRequest).  One lane per v-packet; a lane that finishes takes the next item
// of the launch-wide list (a wave reserves VQ_RESERVE items per atomic), so all 64 lanes of every wave trace until the list is
// empty -- whatever the lengths of the v-packets.  Launch of the v-packet: trace_vpacket_volley (virtual_packet.py:248-386), the
// same operations as the pooled volleys of propagate_wave_kernel; one shell crossing per iteration: vp_shell_step.
template <bool FULL>
__global__ void __launch_bounds__(64) vpacket_trace_kernel(const WaveCold *__restrict__ W)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *geo = reinterpret_cast<double *>(lds_raw);  // r_inner | r_outer | n_e | tau row sums (v-packet screening)
    const GroupArgs &P = W->P;
    const int S = P.n_shells;
    const int lane = threadIdx.x;
    for (int s = lane; s < S; s += 64) {
        geo[s] = glob(P.r_inner)[s]; geo[S + s] = glob(P.r_outer)[s]; geo[2 * S + s] = glob(P.n_e)[s];
        geo[3 * S + s] = P.tau_rowsum ? glob(P.tau_rowsum)[s] : 0.0;
    }
    __syncthreads();
    const unsigned n_items = glob(W->vq_count)[0];
    const int n_v = (int)P.n_vpackets;
    const double t = P.t_exp;
    const double r_in0 = geo[0];
    unsigned res_next = 0, res_end = 0;  // wave-uniform: reserved items not yet handed out
    bool exhausted = n_items == 0;
    bool tracing = false;
    unsigned my_slot = 0;
    int my_sl = 0, w_q = 0, w_used = 0, w_avail = 0;
    unsigned my_visits = 0;
    VpState vs;
    double v_rcp_nu = 0.0;
    bool v_fast = false;
    vs.r = vs.mu = vs.nu = vs.energy = vs.tau = vs.mu0 = 0.0; vs.shell = 0; vs.next_line = 0;
    // screening (tau_prefix.hpp), as in the pooled volleys: an item predicted to be dropped is first traced on the prefix sums
    bool screening = false;
    double v_margin = 0.0, v0_r = 0.0, v0_energy = 0.0;
    int v0_shell = 0, v0_line = 0;
    for (;;) {
        const unsigned long long free_l = __ballot(!tracing);
        bool take = false;
        unsigned my_item = 0;
        if (free_l) {
            if (res_next == res_end && !exhausted) {
                unsigned b = 0;
                if (lane == 0) b = gatomic_add_u32(W->vq_count + 1, (unsigned)VQ_RESERVE);
                b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= n_items) exhausted = true;
                else { res_next = b; res_end = min(b + (unsigned)VQ_RESERVE, n_items); }
            }
            const int n_take = min(__popcll(free_l), (int)(res_end - res_next));
            const int rank = __popcll(free_l & ((1ull << lane) - 1ull));
            take = !tracing && rank < n_take;
            my_item = res_next + (unsigned)rank;
            res_next += (unsigned)n_take;
        }
        if (!__ballot(tracing || take)) {
            if (exhausted) break;
            continue;
        }
        if (take) {
            const unsigned it = glob(W->vq_items)[my_item];
            my_slot = it >> 3; my_sl = (int)(it & 7u);
            const MC_G VolleyRequest *rq = glob(W->vq_req + my_slot);
            const double f_r = rq->r, f_mu = rq->mu, f_nu = rq->nu, f_energy = rq->energy;
            const int f_vdone = rq->vdone;
            const unsigned f_pred = rq->pred_bits;
            // the frame of the volley (:262-300), as the propagation kernel computes it for its pooled volleys
            double mu_min = 0.0, beta_inner = 0.0;
            bool on_inner = false;
            if (f_r > r_in0) {
                const double r_inner_over_r = r_in0 / f_r;
                mu_min = -sqrt(1 - r_inner_over_r * r_inner_over_r);
                if (FULL) mu_min = aberration_lf_to_cmf(f_r, t, mu_min);
            } else {
                on_inner = true;
                if (FULL) {
                    const double inv_c = 1 / C_LIGHT;
                    const double inv_t = 1 / t;
                    beta_inner = r_in0 * inv_t * inv_c;
                }
            }
            const double mu_bin = (1.0 - mu_min) / (double)n_v;
            const double r_dop = doppler_factor<FULL>(f_r / t, f_mu);
            const int i = f_vdone + my_sl;  // index of the v-packet in its volley
            const unsigned before = (f_pred >> f_vdone) & ((1u << my_sl) - 1u);  // predicted roulette draws of the round's earlier v-packets
            const int q = my_sl + __popc(before);
            const double xi = rq->draws[q];
            w_q = q + 1; w_used = 0; w_avail = rq->cnt;
            double v_mu = mu_min + (double)i * mu_bin + xi * mu_bin;
            double weight;
            if (on_inner) {
                if (!FULL) weight = 2 * v_mu / (double)n_v;
                else weight = 2 * (v_mu + beta_inner) / (2 * beta_inner + 1) / (double)n_v;
            } else
                weight = (1 - mu_min) / (double)(2 * n_v);
            if (FULL) v_mu = aberration_cmf_to_lf(f_r, t, v_mu);
            const double v_dop = doppler_factor<FULL>(f_r / t, v_mu);
            const double ratio = r_dop / v_dop;
            vs.r = f_r; vs.mu = v_mu; vs.mu0 = v_mu;  // the log records the (aberrated) launch direction (:337-340,375)
            vs.nu = f_nu * ratio;
            v_rcp_nu = 1.0 / vs.nu;
            v_fast = mid_range(vs.nu);
            vs.energy = f_energy * weight * ratio;
            vs.tau = 0.0; vs.shell = rq->shell; vs.next_line = rq->next_line;
            my_visits = 0;
            tracing = true;
            screening = P.tau_pfx != nullptr && ((f_pred >> i) & 1u) != 0u;
            v_margin = 0.0; v0_r = f_r; v0_energy = vs.energy; v0_shell = vs.shell; v0_line = vs.next_line;
        }
        if (tracing) {
            const MC_G double *dr = glob(W->vq_req + my_slot)->draws;
            auto wdraw = [&]() {
                const double d = dr[w_q + w_used];
                ++w_used;
                return d;
            };
            int draws_left = w_avail - (w_q + w_used);
            int st;
            if (screening) {
                st = vp_screen_step<FULL>(P, wdraw, draws_left, vs, v_margin, v_rcp_nu, v_fast, geo, my_visits);
                if (st == 2) {  // not decided on the prefix sums: again, line by line
                    vs.r = v0_r; vs.mu = vs.mu0; vs.energy = v0_energy; vs.tau = 0.0; vs.shell = v0_shell; vs.next_line = v0_line;
                    my_visits = 0; w_used = 0; screening = false; st = 0;
                }
            } else
                st = vp_shell_step<FULL>(P, wdraw, draws_left, vs, v_rcp_nu, v_fast, geo, my_visits);
            if (st != 0) {
                VpResult r;
                r.nu = vs.nu; r.energy = st == 1 ? vs.energy * mcm::exp(-vs.tau) : 0.0; r.mu0 = vs.mu0;
                r.used = w_used; r.visits = (int)my_visits; r.err = st < 0 ? st : 0; r.pad = 0;
                gstore(W->vp_scratch + ((size_t)my_slot * VP_ROUND + my_sl), r);
                tracing = false;
            }
        }
    }
}

}  // namespace mc
