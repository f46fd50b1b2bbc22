// propagate_wave.hpp -- wave-owner propagation kernel (variant 2): every LANE owns a packet, the wave's G-lane groups
// are a pool of line-sweep workers.
//
// Why (measured on the group kernel, profiles/r01_*): with one packet per G-lane group, (a) the packet's scalar event
// code (boundary distance, tau_event, move, scatter, macro atom) is executed redundantly by all G lanes, i.e. a
// 64-lane wave retires 64/G events per pass through ~700 instructions, and (b) the wave's groups sweep in lockstep, so
// every trace costs the MAXIMUM number of G-line steps over the wave's groups (~2.3x the mean for the exponential-like
// sweep lengths of a Sobolev line list).  Here the two kinds of work are separated:
//
//   event phase  -- lane-per-packet: all 64 lanes run the scalar event code for 64 different packets at once
//                   (epilogue of the finished trace, macro atom, packet hand-over, prologue of the next trace).
//   sweep phase  -- the 64/G groups take the prepared traces from the wave's pool one after the other: a group that
//                   finds its stopping line immediately continues with the next waiting packet, whose first lines
//                   were prefetched while the previous sweep was still running.  All groups stay busy until the pool
//                   is empty, so the sweep cost follows the MEAN sweep length.
//
// The arithmetic of a trace is the group kernel's (same operation order, same serial optical-depth scan, same deferred
// estimator atomics), so per-packet results stay bit-identical to the CPU oracle.  Sweep parameters travel from the
// owner lane to the worker group through LDS (written once in the prologue) and two cross-lane reads; the stopping
// line / distance come back through LDS.  MT19937: the packet's state stays in global memory (seeded by
// seed_states_kernel) and is advanced 8 words at a time by 8-lane subgroups with coalesced accesses; the tempered
// doubles are parked in a per-lane LDS ring from which the lane-per-packet code pops its draws.
#pragma once
#include "mc_device.hpp"
#include "propagate_group.hpp"

namespace mc {

constexpr int WV_RING = 8;  // look-ahead doubles per packet (power of two, >= 8: a refill adds 4)
enum : int { WS_NEED_PACKET = 0, WS_NEED_TRACE = 1, WS_POOL = 2, WS_ASSIGNED = 3, WS_READY = 4, WS_DONE = 5 };
constexpr int RES_PENDING = -1;

// per-wave LDS, structure of arrays indexed by lane
struct WaveShared {
    double comov_nu[64], chi[64], rcp_nu[64], rcp_chi[64], tau_event[64], mur[64];
    double d_boundary[64];  // in: boundary distance of the prepared trace; out: distance of the event found
    double ring[WV_RING][64];
    int res_info[64], res_line[64];
};
struct WaveTracker {  // packets/trackers/tracker_last_interaction.py:8-254; nu/energy/after_nu/after_energy are the
    double radius[64], before_nu[64], before_mu[64], before_energy[64], after_mu[64];  // packet's final nu and energy
    int shell_id[64], interaction_type[64], line_absorb_id[64], line_emit_id[64], interactions_count[64], boundary_buffer[64];
};

template <bool TRACK>
__host__ __device__ constexpr size_t wave_kernel_lds_bytes(int n_shells)
{
    return sizeof(WaveShared) + (TRACK ? sizeof(WaveTracker) : 0) + 2 * (size_t)n_shells * sizeof(double);
}

// One worker slot of a group: the trace it is sweeping (group-uniform values) and this lane's line of the current chunk.
struct SweepSlot {
    int owner;  // lane (in this wave) of the packet being traced, -1: idle
    double nu, energy, mur, r, mu;
    double comov_nu, chi, rcp_nu, rcp_chi, tau_event, d_boundary;
    double tau_carry, d_cont_carry;
    double nu_line, tau_line;
    int cur0;
    unsigned row;
    bool fast;
    bool pend_valid;
    unsigned pend_idx;
    double pend_e, pend_jb;
};

// One G-line step of the line sweep of trace_packet (modes/homologous_rad_packet_transport.py:100-172); the loop body of
// sweep_lines() in propagate_group.hpp with the loop turned inside out.  On a stop (or when the list is exhausted) the
// result is handed to the owner lane through LDS and the slot becomes idle.
template <bool FULL, int G, bool FAST>
__device__ __forceinline__ void sweep_step(const GroupArgs &P, SweepSlot &s, const int j, double *__restrict__ jb,
                                           double *__restrict__ ed, WaveShared &sh, unsigned long long &visits)
{
    const int L = P.n_lines;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : ((G == 8) ? 0xffull : 0xfull);
    if (s.pend_valid && !(P.debug_flags & 1)) {
        atomic_add_f64(&jb[s.pend_idx], s.pend_jb);
        atomic_add_f64(&ed[s.pend_idx], s.pend_e);
    }
    s.pend_valid = false;
    int info = 0, res_line = 0;
    double distance = 0.0;
    bool finished = false;
    if (s.cur0 < L) {
        const int line = s.cur0 + j;
        const bool in_range = line < L;
        // prefetch the next chunk
        const int nline = line + G;
        const bool nin = nline < L;
        const double nu_next = nin ? P.nu_line[(unsigned)nline] : 0.0;
        const double tau_next = nin ? P.tau_t[s.row + (unsigned)nline] : 0.0;

        const double nu_line = s.nu_line, tau_line = s.tau_line;
        const double tau_incl = serial_prefix<G>(s.tau_carry, tau_line, j);
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
        const int first = stop_mask ? __builtin_ctz(stop_mask) : G;
        const int code = stop_b ? 1 : (stop_e ? 2 : (stop_l ? 3 : ((in_range && err) ? 4 : 0)));
        const int first_code = gbcast<G>(code, first & (G - 1));
        const bool visited = in_range && (j < first || (j == first && first_code == 3));
        double pend_e;
        if (!FULL) pend_e = s.energy * (1.0 - exact_div<FAST>(d_trace + s.mur, P.tc, P.rcp_tc));
        else pend_e = s.energy;
        s.pend_valid = visited;
        s.pend_idx = s.row + (unsigned)line;
        s.pend_e = pend_e;
        s.pend_jb = exact_div<FAST>(pend_e, s.nu, s.rcp_nu);
        if (first < G) {
            visits += (unsigned long long)(first + 1);
            finished = true;
            info = first_code;
            res_line = s.cur0 + first;
            if (first_code == 1) distance = s.d_boundary;
            else if (first_code == 2) distance = gbcast<G>(d_cont, first);
            else distance = gbcast<G>(d_trace, first & (G - 1));
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
        finished = true;
        if (s.d_cont_carry < s.d_boundary) { distance = s.d_cont_carry; info = 2 | 8; }
        else { distance = s.d_boundary; info = 1 | 8; }
    }
    if (finished) {
        if (s.pend_valid && !(P.debug_flags & 1)) {
            atomic_add_f64(&jb[s.pend_idx], s.pend_jb);
            atomic_add_f64(&ed[s.pend_idx], s.pend_e);
        }
        s.pend_valid = false;
        if (j == 0) {
            sh.d_boundary[s.owner] = distance;
            sh.res_line[s.owner] = res_line;
            sh.res_info[s.owner] = info;
        }
        s.owner = -1;
    }
}

template <bool FULL, bool TRACK, int G>
__global__ void __launch_bounds__(64) propagate_wave_kernel(GroupArgs P, uint32_t *__restrict__ seeded_states, long long chunk_first,
                                                            long long chunk_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    WaveShared &sh = *reinterpret_cast<WaveShared *>(lds_raw);
    WaveTracker &trk = *reinterpret_cast<WaveTracker *>(lds_raw + sizeof(WaveShared));
    double *lds_J = reinterpret_cast<double *>(lds_raw + sizeof(WaveShared) + (TRACK ? sizeof(WaveTracker) : 0));
    double *lds_nubar = lds_J + P.n_shells;
    const int lane = threadIdx.x;  // one wave per workgroup
    for (int s = lane; s < 2 * P.n_shells; s += 64) lds_J[s] = 0.0;

    const int j = lane & (G - 1);
    const int group_lane0 = lane & ~(G - 1);
    const int copy = P.n_est_copies > 1 ? (xcc_id() % P.n_est_copies) : 0;
    double *jb = P.jblue_t + (size_t)copy * P.est_copy_stride;
    double *ed = P.edot_t + (size_t)copy * P.est_copy_stride;
    const double t = P.t_exp;
    const int L = P.n_lines;

    // ---- owner-lane state: this lane's packet
    Packet p;
    p.r = p.mu = p.nu = p.energy = 0.0; p.shell = 0; p.next_line_id = 0; p.status = ST_IN_PROCESS;
    int state = WS_NEED_PACKET;
    int pkt = 0;          // index of the packet in this launch's chunk
    double dop = 1.0;     // Doppler factor at the start of the prepared trace
    int pflags = 0;       // bit 0: exact-division fast path is safe; bits 1-2: delta_shell + 1
    int r_gpos = 0, r_head = 0, r_cnt = 0;  // MT19937: next state block to regenerate; LDS ring of tempered doubles
    unsigned draws = 0, events = 0, macro = 0;
    bool exhausted = false;  // wave-uniform: the chunk has no more packets
    // ---- worker state of this lane's group
    SweepSlot cur;
    cur.owner = -1; cur.pend_valid = false; cur.cur0 = 0; cur.row = 0; cur.fast = true;
    cur.nu = cur.energy = cur.mur = cur.r = cur.mu = cur.comov_nu = cur.chi = cur.rcp_nu = cur.rcp_chi = 0.0;
    cur.tau_event = cur.d_boundary = cur.tau_carry = cur.d_cont_carry = cur.nu_line = cur.tau_line = 0.0;
    cur.pend_idx = 0; cur.pend_e = cur.pend_jb = 0.0;
    int nxt_owner = -1, n_cursor = 0;
    unsigned n_row = 0;
    double n_nu = 0.0, n_tau = 0.0;
    unsigned long long visits = 0;

    auto draw = [&]() {
        const double v = sh.ring[r_head][lane];
        r_head = (r_head + 1) & (WV_RING - 1);
        --r_cnt;
        ++draws;
        return v;
    };
    // wave-uniform: give every lane of `need` four more doubles (8 stream words, regenerated in place by an 8-lane
    // subgroup with coalesced accesses; 624 = 78 * 8, so a block never wraps)
    auto refill = [&](unsigned long long need) {
        const int sj = lane & 7;
        while (need) {
            int my_owner = -1;
            unsigned long long rest = need;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (rest) {
                    const int o = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    if ((lane >> 3) == q) my_owner = o;
                }
            }
            const unsigned long long served = need & ~rest;
            need = rest;
            const int src = my_owner >= 0 ? my_owner : lane;
            const int o_pkt = __shfl(pkt, src), o_gpos = __shfl(r_gpos, src), o_tail = __shfl((r_head + r_cnt) & (WV_RING - 1), src);
            if (my_owner >= 0) {
                uint32_t *st = seeded_states + (size_t)o_pkt * MT_N;
                const int k = o_gpos + sj;
                const int k1 = (k + 1 == MT_N) ? 0 : k + 1;
                const int km = (k + 397 >= MT_N) ? k + 397 - MT_N : k + 397;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t a = st[k], b = st[k1], c = st[km];
                const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
                const uint32_t v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                st[k] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint32_t nb = (uint32_t)__shfl_down((int)v, 1, 8);
                if (!(sj & 1)) sh.ring[(o_tail + (sj >> 1)) & (WV_RING - 1)][my_owner] = GroupRng<8>::to_double(v, nb);
            }
            if ((served >> lane) & 1ull) {
                r_cnt += 4;
                r_gpos = (r_gpos + 8 == MT_N) ? 0 : r_gpos + 8;
            }
        }
    };

    for (;;) {
        // ============================================================ event phase (lane-per-packet)
        // ---- epilogue of the finished traces: move, estimators, boundary / scattering
        refill(__ballot(state == WS_READY && r_cnt < 2));
        int err = 0, type = 0, emit = -1, absorb = -1, mb0 = 0, mb1 = 0;
        double inv_new = 1.0, before_nu = 0.0, before_mu = 0.0, before_energy = 0.0, scat_comov_nu = 0.0;
        bool in_macro = false, interacted = false;
        if (state == WS_READY) {
            const int info = sh.res_info[lane];
            const double distance = sh.d_boundary[lane];
            const int code = info & 7;
            if (!(info & 8)) p.next_line_id = sh.res_line[lane];
            if (code == 4) err = ERR_MONTECARLO;
            type = code == 1 ? IT_BOUNDARY : (code == 2 ? IT_ESCATTERING : IT_LINE);
            if (!err) {
                // move_r_packet + update_estimators_bulk (packets/movement.py:31-76)
                const double r = p.r;
                if (distance > 0.0) {
                    const double new_r = sqrt(r * r + distance * distance + 2.0 * r * distance * p.mu);
                    const double mu_new = (p.mu * r + distance) / new_r;
                    const double comov_nu = p.nu * dop;
                    const double comov_energy = p.energy * dop;
                    const double dist_est = FULL ? distance * dop : distance;
                    if (!(P.debug_flags & 2)) {
                        atomicAdd(&lds_J[p.shell], comov_energy * dist_est);
                        atomicAdd(&lds_nubar[p.shell], comov_energy * dist_est * comov_nu);
                    }
                    p.mu = mu_new;
                    p.r = new_r;
                }
                if (type == IT_BOUNDARY) {
                    if (TRACK) trk.boundary_buffer[lane] += 1;
                    cross_shell(p.shell, p.status, ((pflags >> 1) & 3) - 1, P.n_shells);
                } else {
                    interacted = true;
                    before_nu = p.nu; before_mu = p.mu; before_energy = p.energy;
                    absorb = (type == IT_LINE) ? p.next_line_id : -1;
                    // common part of line_scatter_event (interaction_event_callers.py:187-239) and thomson_scatter
                    // (interaction_events.py:184-217): Doppler with the old angle, new isotropic angle, Doppler back
                    const double vel = p.r / t;
                    const double old_dop = doppler_factor<FULL>(vel, p.mu);
                    scat_comov_nu = p.nu * old_dop;
                    const double comov_energy = p.energy * old_dop;
                    p.mu = 2.0 * draw() - 1.0;
                    inv_new = inverse_doppler_factor<FULL>(vel, p.mu);
                    p.energy = comov_energy * inv_new;
                    if (type == IT_LINE) {
                        emit = p.next_line_id;
                        if (P.line_interaction_type != 0) {
                            const int2 blk = P.line_block[(unsigned)p.next_line_id];
                            mb0 = blk.x; mb1 = blk.y;
                            in_macro = true;
                        }
                    }
                }
            }
        }
        // ---- macro_atom_interaction (macro_atom.py:52-104), one jump per pass, lane-per-packet
        while (__ballot(in_macro)) {
            refill(__ballot(in_macro && r_cnt < 1));
            if (in_macro) {
                const unsigned row = (unsigned)p.shell * (unsigned)P.n_trans;
                const double event = draw();
                double carry = 0.0;
                int k = mb0;
                bool found = false;
                for (; k < mb1; ++k) {
                    carry += P.prob_t[row + (unsigned)k];
                    if (carry > event) { found = true; break; }
                }
                if (!found) { macro += (unsigned)(mb1 - mb0); err = ERR_MACRO_ATOM; in_macro = false; }
                else {
                    macro += (unsigned)(k - mb0 + 1);
                    const int4 rec = P.trans_rec[(unsigned)k];
                    emit = rec.x; mb0 = rec.z; mb1 = rec.w;
                    if (rec.y < 0) {
                        in_macro = false;
                        if (rec.y != -1) err = ERR_UNSUPPORTED;
                    }
                }
            }
        }
        // ---- finish the interaction, hand finished packets over
        if (state == WS_READY) {
            if (interacted && !err) {
                int emit_id = -1;
                if (type == IT_LINE) {  // line_emission (interaction_events.py:227-258); its inverse Doppler factor == inv_new
                    p.nu = P.nu_line[emit] * inv_new;
                    p.next_line_id = emit + 1;
                    emit_id = emit;
                } else {
                    p.nu = scat_comov_nu * inv_new;
                }
                if (FULL) p.mu = aberration_cmf_to_lf(p.r, t, p.mu);
                if (TRACK) {
                    trk.before_nu[lane] = before_nu; trk.before_mu[lane] = before_mu; trk.before_energy[lane] = before_energy;
                    trk.line_absorb_id[lane] = absorb; trk.line_emit_id[lane] = emit_id;
                    trk.after_mu[lane] = p.mu;
                    trk.interactions_count[lane] += 1 + trk.boundary_buffer[lane];
                    trk.boundary_buffer[lane] = 0;
                    trk.radius[lane] = p.r; trk.shell_id[lane] = p.shell;
                    trk.interaction_type[lane] = type;
                }
            }
            state = WS_NEED_TRACE;
            if (err || p.status != ST_IN_PROCESS) {
                const long long i = chunk_first + pkt;
                const DeviceProblem *C = P.cold;
                if (err) {
                    atomicMin(&C->first_error[0], i);
                    C->out_nu[i] = (double)err;
                    C->out_e[i] = -99.0;
                } else {
                    // set_packet_collection_output (modes/montecarlo_transport.py:70-90)
                    C->out_nu[i] = p.nu;
                    C->out_e[i] = (p.status == ST_REABSORBED) ? -p.energy : p.energy;
                    if (TRACK) {
                        const bool any = trk.interaction_type[lane] >= 0;
                        const double nan = __builtin_nan("");
                        C->li_radius[i] = any ? trk.radius[lane] : nan;
                        C->li_nu[i] = any ? p.nu : nan;
                        C->li_energy[i] = any ? p.energy : nan;
                        C->li_before_nu[i] = any ? trk.before_nu[lane] : nan;
                        C->li_before_mu[i] = any ? trk.before_mu[lane] : nan;
                        C->li_before_energy[i] = any ? trk.before_energy[lane] : nan;
                        C->li_after_nu[i] = any ? p.nu : nan;
                        C->li_after_mu[i] = any ? trk.after_mu[lane] : nan;
                        C->li_after_energy[i] = any ? p.energy : nan;
                        C->li_shell_id[i] = trk.shell_id[lane]; C->li_interaction_type[i] = trk.interaction_type[lane];
                        C->li_line_absorb_id[i] = trk.line_absorb_id[lane]; C->li_line_emit_id[i] = trk.line_emit_id[lane];
                        C->li_interactions_count[i] = trk.interactions_count[lane];
                    }
                }
                state = WS_NEED_PACKET;
            }
        }
        // ---- fetch packets (one global atomic per wave and pass)
        {
            const unsigned long long need_pkt = __ballot(state == WS_NEED_PACKET);
            if (need_pkt) {
                long long base = chunk_count;
                if (!exhausted) {
                    const int n_want = __popcll(need_pkt);
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(P.next_packet, (unsigned long long)n_want);
                    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b);
                    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
                    base = (long long)(((unsigned long long)bhi << 32) | blo);
                    if (base + n_want >= chunk_count) exhausted = true;
                }
                if (state == WS_NEED_PACKET) {
                    const long long mine = base + __popcll(need_pkt & ((1ull << lane) - 1ull));
                    if (mine >= chunk_count) state = WS_DONE;
                    else {
                        pkt = (int)mine;
                        r_gpos = r_head = r_cnt = 0;
                        const long long i = chunk_first + mine;
                        const DeviceProblem *C = P.cold;
                        p.r = C->r0[i]; p.mu = C->mu0[i]; p.nu = C->nu0[i]; p.energy = C->e0[i];
                        p.shell = 0; p.status = ST_IN_PROCESS;
                        if (TRACK) {
                            trk.shell_id[lane] = -1; trk.interaction_type[lane] = -1; trk.line_absorb_id[lane] = -1;
                            trk.line_emit_id[lane] = -1; trk.interactions_count[lane] = 0;
                            trk.boundary_buffer[lane] = 0;  // -1 + the initial track_boundary_event
                        }
                        {   // set_packet_props_{partial,full}_relativity (classic/packet_propagation.py:254-318)
                            const double velocity = p.r / t;
                            const double inv = inverse_doppler_factor<FULL>(velocity, p.mu);
                            if (FULL) {
                                const double beta = velocity / C_LIGHT;
                                p.nu *= inv; p.energy *= inv;
                                p.mu = (p.mu + beta) / (1 + beta * p.mu);
                            } else { p.nu *= inv; p.energy *= inv; }
                        }
                        {   // initialize_line_id (packets/radiative_packet.py:96-110) through the frequency-bucket index
                            const double velocity = p.r / t;
                            const double comov_nu = p.nu * doppler_factor<FULL>(velocity, p.mu);
                            int lo;
                            const long long kk = (long long)((unsigned long long)__double_as_longlong(comov_nu > 0.0 ? comov_nu : 0.0) >> P.bucket_shift) - P.bucket_kmin;
                            if (kk >= P.bucket_n) lo = 0;
                            else if (kk < 0) lo = L;
                            else {
                                lo = P.bucket_first[kk];
                                const int hi = kk > 0 ? P.bucket_first[kk - 1] : L;
                                while (lo < hi && P.nu_line[(unsigned)lo] >= comov_nu) ++lo;
                            }
                            if (lo == L) lo -= 1;
                            p.next_line_id = lo;
                        }
                        state = WS_NEED_TRACE;
                    }
                }
            }
        }
        if (__ballot(state != WS_DONE) == 0ull) break;
        // ---- prologue of the next trace (trace_packet, modes/homologous_rad_packet_transport.py:30-98)
        refill(__ballot(state == WS_NEED_TRACE && r_cnt < 1));
        if (state == WS_NEED_TRACE) {
            const double velocity = p.r / t;
            dop = doppler_factor<FULL>(velocity, p.mu);
            double chi_e = P.n_e[p.shell] * P.sigma_thomson;
            if (FULL) chi_e *= dop;
            double d_boundary;
            int delta;
            distance_boundary(p.r, p.mu, P.r_inner[p.shell], P.r_outer[p.shell], d_boundary, delta);
            const double tau_event = -mcm::log(draw());
            const double comov_nu = p.nu * dop;
            ++events;
            const bool fast = mid_range(p.nu) && mid_range(chi_e) && mid_range(tau_event) && mid_range(p.energy) &&
                              mid_range(p.r) && mid_range(comov_nu) && mid_range(P.t_exp) && !(P.debug_flags & 4);
            pflags = (fast ? 1 : 0) | ((delta + 1) << 1);
            sh.comov_nu[lane] = comov_nu; sh.chi[lane] = chi_e; sh.tau_event[lane] = tau_event; sh.d_boundary[lane] = d_boundary;
            sh.rcp_nu[lane] = 1.0 / p.nu; sh.rcp_chi[lane] = 1.0 / chi_e; sh.mur[lane] = p.mu * p.r;
            sh.res_info[lane] = RES_PENDING;
            state = WS_POOL;
        }

        // ============================================================ sweep phase (G-lane groups work off the pool)
        for (;;) {
            unsigned long long pool = __ballot(state == WS_POOL);
            const unsigned long long busy = __ballot(cur.owner >= 0 || nxt_owner >= 0);
            if (!pool && !busy) break;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): everything prefetched during the previous step has landed
            // ---- a group whose sweep ended continues with the packet it prefetched
            const bool promote = cur.owner < 0 && nxt_owner >= 0;
            if (__ballot(promote)) {
                const int src = nxt_owner >= 0 ? nxt_owner : lane;
                const double f_nu = __shfl(p.nu, src), f_energy = __shfl(p.energy, src);
                const double f_r = FULL ? __shfl(p.r, src) : 0.0, f_mu = FULL ? __shfl(p.mu, src) : 0.0;
                const int f_flags = __shfl(pflags, src);
                if (promote) {
                    cur.owner = nxt_owner;
                    nxt_owner = -1;
                    cur.nu = f_nu; cur.energy = f_energy; cur.r = f_r; cur.mu = f_mu;
                    cur.fast = (f_flags & 1) != 0;
                    cur.comov_nu = sh.comov_nu[cur.owner]; cur.chi = sh.chi[cur.owner]; cur.tau_event = sh.tau_event[cur.owner];
                    cur.d_boundary = sh.d_boundary[cur.owner]; cur.rcp_nu = sh.rcp_nu[cur.owner]; cur.rcp_chi = sh.rcp_chi[cur.owner];
                    cur.mur = sh.mur[cur.owner];
                    cur.cur0 = n_cursor; cur.row = n_row;
                    cur.nu_line = n_nu; cur.tau_line = n_tau;
                    cur.tau_carry = 0.0;
                    cur.d_cont_carry = cur.fast ? exact_div<true>(cur.tau_event, cur.chi, cur.rcp_chi)
                                                : exact_div<false>(cur.tau_event, cur.chi, cur.rcp_chi);
                    cur.pend_valid = false;
                }
            }
            // ---- hand waiting packets to the groups without a prefetched one and start loading their first lines
            unsigned long long free_groups = __ballot(nxt_owner < 0 && j == 0);
            if (pool && free_groups) {
                bool newly = false;
                while (pool && free_groups) {
                    const int gl = __builtin_ctzll(free_groups);
                    const int o = __builtin_ctzll(pool);
                    free_groups &= free_groups - 1;
                    pool &= pool - 1;
                    if (group_lane0 == gl) { nxt_owner = o; newly = true; }
                    if (lane == o) state = WS_ASSIGNED;
                }
                const int src = nxt_owner >= 0 ? nxt_owner : lane;
                const int f_line = __shfl(p.next_line_id, src), f_shell = __shfl(p.shell, src);
                if (newly) {
                    n_cursor = f_line;
                    n_row = (unsigned)f_shell * (unsigned)L;
                    const int line = n_cursor + j;
                    const bool in = line < L;
                    n_nu = in ? P.nu_line[(unsigned)line] : 0.0;
                    n_tau = in ? P.tau_t[n_row + (unsigned)line] : 0.0;
                }
            }
            // ---- one G-line step of every running sweep
            if (cur.owner >= 0) {
                if (cur.fast) sweep_step<FULL, G, true>(P, cur, j, jb, ed, sh, visits);
                else sweep_step<FULL, G, false>(P, cur, j, jb, ed, sh, visits);
            }
            if (state == WS_ASSIGNED && sh.res_info[lane] != RES_PENDING) state = WS_READY;
        }
    }

    const DeviceProblem *C = P.cold;
    for (int s = lane; s < P.n_shells; s += 64) {
        if (lds_J[s] != 0.0) atomic_add_f64(&C->J[s], lds_J[s]);
        if (lds_nubar[s] != 0.0) atomic_add_f64(&C->nubar[s], lds_nubar[s]);
    }
    // counters: wave-reduce, one atomic each
    unsigned long long v = (j == 0) ? visits : 0ull;  // group-uniform: count once per group
    unsigned long long e = events, m = macro, d = draws;
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off); e += __shfl_down(e, off); m += __shfl_down(m, off); d += __shfl_down(d, off);
    }
    if (lane == 0) {
        atomicAdd(&C->counters[0], v);
        atomicAdd(&C->counters[1], e);
        atomicAdd(&C->counters[2], m);
        atomicAdd(&C->counters[5], d);
    }
}

}  // namespace mc
