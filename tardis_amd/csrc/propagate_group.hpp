// propagate_group.hpp -- cooperative propagation kernel (variant 1): 16 lanes per packet.
//
// Why: the dominant cost of the lane-per-packet kernel is the 2 scattered fp64 atomics per line visit (measured
// ceiling on MI355X: ~24 G random fp64 atomics/s chip-wide, but ~170 G/s when 16 lanes hit 16 consecutive doubles,
// profiles/r01_microbench.txt).  Here a 16-lane group (one DPP row, four groups per wave) owns one packet and sweeps
// the sorted line list 16 lines at a time:
//   * nu_line[cur..cur+15] and tau[shell][cur..cur+15] are two coalesced 128-byte loads,
//   * every lane evaluates "its" line (distance, estimator energy),
//   * the running Sobolev optical depth is carried across lanes IN THE REFERENCE'S SERIAL ORDER (bit-exact),
//   * a 16-bit ballot picks the first line at which the reference's loop would have stopped,
//   * lanes before it issue the j_blue / Edotlu atomics as 128-byte-contiguous groups.
// The packet's scalar event code (boundary distance, tau_event, move, scatter, macro atom) is executed redundantly
// by the 16 lanes, so no cross-lane traffic is needed for it.  The packet's MT19937 state lives in LDS (2496 B per
// group) and is regenerated cooperatively 16 words at a time; raw seeded states are produced by a separate
// lane-per-packet kernel (the init_genrand recurrence is serial per packet).
#pragma once
#include "mc_device.hpp"
#include "propagate_lane.hpp"  // Tracker, xcc_id

namespace mc {

constexpr int GROUP = 16;
constexpr int GROUPS_PER_BLOCK = 16;   // 256 threads
constexpr int PACKET_BATCH = 16;       // packets reserved per global atomic

// ---- seeding kernel: raw init_genrand state, [packet][624] contiguous, one packet per lane.
// Stores go through an LDS tile so that a 16-lane group writes 64 contiguous bytes of one packet's state.
__global__ void __launch_bounds__(256) seed_states_kernel(const uint32_t *__restrict__ seeds, uint32_t *__restrict__ states,
                                                          long long first, long long count)
{
    __shared__ uint32_t tile[256][17];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < count;
    uint32_t x = valid ? seeds[first + i] : 0u;
    const long long block_first = (long long)blockIdx.x * blockDim.x;
    for (int base = 0; base < MT_N; base += 16) {
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            int k = base + w;
            if (k > 0) x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
            tile[threadIdx.x][w] = x;
        }
        __syncthreads();
        // 256 packets x 16 words: thread t writes word (t & 15) of packets (t >> 4) + 16*r
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int pk = (threadIdx.x >> 4) + 16 * r;
            long long gi = block_first + pk;
            if (gi < count && base + (threadIdx.x & 15) < MT_N)
                states[(size_t)gi * MT_N + base + (threadIdx.x & 15)] = tile[pk][threadIdx.x & 15];
        }
        __syncthreads();
    }
}

// ---- MT19937 in LDS, one state per 16-lane group, all lanes of the group call every method together
struct GroupRng {
    uint32_t *mt;    // LDS, 624 words
    int idx;         // next output word (group-uniform)
    int fresh;       // words [0, fresh) of the current generation are already regenerated (group-uniform)
    long long draws;

    __device__ __forceinline__ void regenerate16(int j)
    {   // regenerate words [fresh, fresh+16) in place; fresh is a multiple of 16 and 624 = 39*16
        const int k = fresh + j;
        const int k1 = (k + 1 == MT_N) ? 0 : k + 1;
        const int km = (k + 397 >= MT_N) ? k + 397 - MT_N : k + 397;
        const uint32_t a = mt[k], b = mt[k1], c = mt[km];
        uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        uint32_t v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        // word 623 needs the NEW word 0 (already regenerated); every other k+1 must be the OLD value, which is
        // guaranteed because all 16 reads above are issued before the 16 writes below.
        mt[k] = v;
        fresh += 16;
    }
    __device__ __forceinline__ uint32_t next_u32(int j)
    {
        if (idx == MT_N) { idx = 0; fresh = 0; }
        if (idx >= fresh) regenerate16(j);
        uint32_t v = mt[idx++];
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
    __device__ __forceinline__ double random(int j)
    {
        uint32_t a = next_u32(j) >> 5, b = next_u32(j) >> 6;
        ++draws;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

__device__ __forceinline__ double group_bcast(double v, int src) { return __shfl(v, src, GROUP); }
__device__ __forceinline__ int group_bcast(int v, int src) { return __shfl(v, src, GROUP); }

struct GroupCounters { unsigned long long visits = 0, events = 0, macro = 0; };

// ---- one cooperative trace_packet (modes/homologous_rad_packet_transport.py:30-174)
// All arguments / results are group-uniform except the lane index j.  Returns 0 or a negative error code.
template <bool FULL>
__device__ __forceinline__ int trace_packet_group(const DeviceProblem &P, Packet &p, GroupRng &rng, const int j,
                                                  const double chi_cont, double *__restrict__ jb, double *__restrict__ ed,
                                                  double &distance, int &type, int &delta_shell, GroupCounters &cn)
{
    const int L = P.n_lines;
    const double t = P.t_exp;
    const int start = p.next_line_id;
    const double *__restrict__ tau_row = P.tau_t + (size_t)p.shell * L;
    // software prefetch of the first chunk: the loads fly while the scalar prologue (sqrt, log) runs
    int line = start + j;
    bool in_range = line < L;
    double nu_line = in_range ? P.nu_line[line] : 0.0;
    double tau_line = in_range ? tau_row[line] : 0.0;

    double d_boundary;
    distance_boundary(p.r, p.mu, P.r_inner[p.shell], P.r_outer[p.shell], d_boundary, delta_shell);
    const double tau_event = -mcm::log(rng.random(j));
    const double velocity = p.r / t;
    const double dop = doppler_factor<FULL>(velocity, p.mu);
    const double comov_nu = p.nu * dop;
    const double mur = p.mu * p.r;
    const double tc = t * C_LIGHT;
    double *__restrict__ jb_row = jb + (size_t)p.shell * L;
    double *__restrict__ ed_row = ed + (size_t)p.shell * L;
    const int last = L - 1;
    const int lane = threadIdx.x & 63;
    const int gshift = lane & 48;
    double tau_carry = 0.0;               // tau_trace_line_combined before the first line of this chunk
    double d_cont_carry = tau_event / chi_cont;  // distance_continuous in force at the first line of this chunk
    cn.events++;

    for (int cur0 = start; cur0 < L; cur0 += GROUP) {
        if (cur0 != start) {  // later chunks: plain loads (prefetched only for the first one)
            line = cur0 + j;
            in_range = line < L;
            nu_line = in_range ? P.nu_line[line] : 0.0;
            tau_line = in_range ? tau_row[line] : 0.0;
        }
        // --- serial-order inclusive prefix of tau over the chunk: ((carry + t0) + t1) + ... + tj
        double tau_incl = tau_carry;
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            double ti = group_bcast(tau_line, i);
            tau_incl = tau_incl + ((j >= i) ? ti : 0.0);
        }
        double tau_prev = __shfl_up(tau_incl, 1, GROUP);  // tau_trace_line_combined before this lane's line
        // distance_continuous in force when this lane's line is examined
        const double d_cont = (j == 0) ? d_cont_carry : (tau_event - tau_prev) / chi_cont;
        // --- this lane's line
        double d_trace = 0.0;
        bool err = false;
        if (in_range) err = !distance_line<FULL>(p.nu, p.r, p.mu, comov_nu, line == last, nu_line, t, d_trace);
        const double tau_combined = tau_incl + chi_cont * d_trace;
        double dmin = d_trace;  // Python min(d_trace, d_boundary, d_cont)
        if (d_boundary < dmin) dmin = d_boundary;
        if (d_cont < dmin) dmin = d_cont;
        const bool stop_b = in_range && !err && d_trace != 0 && dmin == d_boundary;
        const bool stop_e = in_range && !err && d_trace != 0 && !stop_b && dmin == d_cont;
        const bool stop_l = in_range && !err && !stop_b && !stop_e && tau_combined > tau_event && !P.disable_line_scattering;
        const bool stop = stop_b || stop_e || stop_l || (in_range && err);
        const unsigned stop_mask = (unsigned)((__ballot(stop) >> gshift) & 0xffffull);
        const int first = stop_mask ? __builtin_ctz(stop_mask) : GROUP;  // group-uniform
        // lines before the stopping one are passed (estimators updated); a LINE stop updates its own line too
        const int first_l = group_bcast((int)stop_l, first & 15);
        const bool first_is_line = (first < GROUP) && first_l;
        const bool visited = in_range && (j < first || (j == first && first_is_line));
        if (visited && !(P.debug_flags & 1)) {
            double energy;
            if (!FULL) energy = p.energy * (1.0 - ((d_trace + mur) / tc));
            else energy = p.energy;
            atomic_add_f64(&jb_row[line], energy / p.nu);
            atomic_add_f64(&ed_row[line], energy);
        }
        if (first < GROUP) {
            const int n_in = min(first + 1, L - cur0);
            cn.visits += (unsigned long long)n_in;
            if (group_bcast((int)err, first)) return ERR_MONTECARLO;
            p.next_line_id = cur0 + first;
            const int sb = group_bcast((int)stop_b, first), se = group_bcast((int)stop_e, first);
            if (sb) { type = IT_BOUNDARY; distance = d_boundary; }
            else if (se) { type = IT_ESCATTERING; distance = group_bcast(d_cont, first); }
            else { type = IT_LINE; distance = group_bcast(d_trace, first); }
            return 0;
        }
        // whole chunk passed: carry the running optical depth and the continuum distance into the next chunk
        const int n_in = min(GROUP, L - cur0);
        cn.visits += (unsigned long long)n_in;
        tau_carry = group_bcast(tau_incl, n_in - 1);
        d_cont_carry = (tau_event - tau_carry) / chi_cont;
    }
    // for-else (lines 157-172): the line list is exhausted; next_line_id is left untouched
    if (d_cont_carry < d_boundary) { distance = d_cont_carry; type = IT_ESCATTERING; }
    else { distance = d_boundary; type = IT_BOUNDARY; }
    return 0;
}

// macro_atom_interaction (macro_atom.py:52-104); scalar walk executed redundantly by the group (block sizes are small)
__device__ inline int macro_atom_group(const DeviceProblem &P, GroupRng &rng, const int j, int level, int shell, int &out_line,
                                       int &out_type, GroupCounters &cn)
{
    const double *prob_row = P.prob_t + (size_t)shell * P.n_trans;
    int ttype = 0, tid = -1;
    while (ttype >= 0) {
        double probability = 0.0;
        double event = rng.random(j);
        int b0 = P.block_edge[level], b1 = P.block_edge[level + 1];
        bool found = false;
        for (tid = b0; tid < b1; ++tid) {
            cn.macro++;
            probability += prob_row[tid];
            if (probability > event) {
                level = P.dest[tid];
                ttype = P.ttype[tid];
                found = true;
                break;
            }
        }
        if (!found) return ERR_MACRO_ATOM;
    }
    out_line = P.tline[tid];
    out_type = ttype;
    return 0;
}

template <bool FULL, bool TRACK>
__global__ void __launch_bounds__(256) propagate_group_kernel(DeviceProblem P, const uint32_t *__restrict__ seeded_states,
                                                              long long chunk_first, long long chunk_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t *lds_mt = reinterpret_cast<uint32_t *>(lds_raw);                       // [16 groups][624]
    double *lds_J = reinterpret_cast<double *>(lds_raw + GROUPS_PER_BLOCK * MT_N * 4);  // [S]
    double *lds_nubar = lds_J + P.n_shells;
    for (int s = threadIdx.x; s < 2 * P.n_shells; s += blockDim.x) lds_J[s] = 0.0;
    __syncthreads();

    const int j = threadIdx.x & (GROUP - 1);
    const int g = threadIdx.x >> 4;
    const int copy = P.n_est_copies > 1 ? (xcc_id() % P.n_est_copies) : 0;
    double *jb = P.jblue_t + (size_t)copy * P.est_copy_stride;
    double *ed = P.edot_t + (size_t)copy * P.est_copy_stride;
    const double t = P.t_exp;

    GroupRng rng;
    rng.mt = lds_mt + g * MT_N;
    rng.idx = 0; rng.fresh = 0; rng.draws = 0;
    GroupCounters cn;
    unsigned long long draws_total = 0;

    // wave-level packet batches: PACKET_BATCH indices are reserved per global atomic and handed to the wave's groups.
    // batch_next / batch_end / exhausted are wave-uniform and only modified in wave-uniform control flow.
    long long batch_next = 0, batch_end = 0;
    bool exhausted = false;
    const int lane = threadIdx.x & 63;
    Packet p;
    p.status = ST_EMITTED;  // "needs a packet"
    p.r = p.mu = p.nu = p.energy = 0.0; p.shell = 0; p.next_line_id = 0;
    long long pkt = -1;
    bool done = false;      // this group has no more work
    Tracker trk;
    if (TRACK) trk.init();

    for (;;) {
        // ---------------------------------------------------------------- fetch packets (wave-uniform step)
        const bool need = !done && p.status != ST_IN_PROCESS;
        const unsigned long long need_leaders = __ballot(need && j == 0);
        if (need_leaders) {
            const int n_want = __popcll(need_leaders);
            const int rank = __popcll(need_leaders & ((1ull << lane) - 1ull));
            long long mine = -1;
            int served = 0;
            while (served < n_want) {
                if (batch_next == batch_end) {
                    if (exhausted) break;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(P.next_packet, (unsigned long long)PACKET_BATCH);
                    base = __shfl(base, 0, 64);
                    batch_next = min((long long)base, chunk_count);
                    batch_end = min((long long)base + PACKET_BATCH, chunk_count);
                    if (batch_next >= batch_end) { exhausted = true; break; }
                }
                const int take = (int)min((long long)(n_want - served), batch_end - batch_next);
                if (need && j == 0 && rank >= served && rank < served + take) mine = batch_next + (rank - served);
                batch_next += take;
                served += take;
            }
            mine = __shfl(mine, 0, GROUP);
            if (need) {
                if (mine < 0) done = true;
                else {
                    pkt = mine;
                    // load the seeded MT state (2496 contiguous bytes) into LDS: 39 coalesced 64-byte group loads
                    const uint32_t *src = seeded_states + (size_t)pkt * MT_N;
                    for (int k = j; k < MT_N; k += GROUP) rng.mt[k] = src[k];
                    draws_total += (unsigned long long)rng.draws;
                    rng.idx = 0; rng.fresh = 0; rng.draws = 0;
                    const long long i = chunk_first + pkt;
                    p.r = P.r0[i]; p.mu = P.mu0[i]; p.nu = P.nu0[i]; p.energy = P.e0[i];
                    p.shell = 0; p.status = ST_IN_PROCESS;
                    if (TRACK) trk.init();
                    {   // set_packet_props_{partial,full}_relativity (classic/packet_propagation.py:254-318)
                        double velocity = p.r / t;
                        double inv = inverse_doppler_factor<FULL>(velocity, p.mu);
                        if (FULL) {
                            double beta = (p.r / t) / C_LIGHT;
                            p.nu *= inv; p.energy *= inv;
                            p.mu = (p.mu + beta) / (1 + beta * p.mu);
                        } else { p.nu *= inv; p.energy *= inv; }
                    }
                    {   // initialize_line_id (packets/radiative_packet.py:96-110)
                        double velocity = p.r / t;
                        double comov_nu = p.nu * doppler_factor<FULL>(velocity, p.mu);
                        int lo = 0, hi = P.n_lines;
                        while (lo < hi) {
                            int mid = (lo + hi) >> 1;
                            if (P.nu_line[mid] >= comov_nu) lo = mid + 1; else hi = mid;
                        }
                        if (lo == P.n_lines) lo -= 1;
                        p.next_line_id = lo;
                    }
                    if (TRACK) trk.boundary_buffer += 1;
                }
            }
        }
        if (__ballot(!done) == 0ull) break;
        if (done) continue;

        // ---------------------------------------------------------------- one event of this group's packet
        double velocity = p.r / t;
        double dop = doppler_factor<FULL>(velocity, p.mu);
        double chi_e = P.n_e[p.shell] * P.sigma_thomson;
        if (FULL) chi_e *= dop;
        double distance;
        int type = 0, delta = 0;
        int err = trace_packet_group<FULL>(P, p, rng, j, chi_e, jb, ed, distance, type, delta, cn);
        if (!err) {
            // move_r_packet + update_estimators_bulk (packets/movement.py:31-76)
            double r = p.r;
            if (distance > 0.0) {
                double new_r = sqrt(r * r + distance * distance + 2.0 * r * distance * p.mu);
                double mu_new = (p.mu * r + distance) / new_r;
                double comov_nu = p.nu * dop;
                double comov_energy = p.energy * dop;
                double dist_est = FULL ? distance * dop : distance;
                if (j == 0 && !(P.debug_flags & 2)) {
                    atomicAdd(&lds_J[p.shell], comov_energy * dist_est);
                    atomicAdd(&lds_nubar[p.shell], comov_energy * dist_est * comov_nu);
                }
                p.mu = mu_new;
                p.r = new_r;
            }
            if (type == IT_BOUNDARY) {
                if (TRACK) trk.boundary_buffer += 1;
                cross_shell(p.shell, p.status, delta, P.n_shells);
            } else if (type == IT_LINE) {
                if (TRACK) {
                    trk.before_nu = p.nu; trk.before_mu = p.mu; trk.before_energy = p.energy;
                    trk.line_absorb_id = p.next_line_id;
                }
                // line_scatter_event (interaction_event_callers.py:187-239)
                double vel = p.r / t;
                double old_dop = doppler_factor<FULL>(vel, p.mu);
                p.mu = 2.0 * rng.random(j) - 1.0;
                double inv_new = inverse_doppler_factor<FULL>(vel, p.mu);
                double comov_energy = p.energy * old_dop;
                p.energy = comov_energy * inv_new;
                int emit = p.next_line_id;
                if (P.line_interaction_type != 0) {
                    double comov_nu = p.nu * old_dop;
                    p.nu = comov_nu * inv_new;
                    int ttype;
                    err = macro_atom_group(P, rng, j, P.line2level[p.next_line_id], p.shell, emit, ttype, cn);
                    if (!err && ttype != -1) err = ERR_UNSUPPORTED;
                }
                if (!err) {
                    // line_emission (interaction_events.py:227-258)
                    double inv = inverse_doppler_factor<FULL>(p.r / t, p.mu);
                    p.nu = P.nu_line[emit] * inv;
                    p.next_line_id = emit + 1;
                    if (FULL) p.mu = aberration_cmf_to_lf(p.r, t, p.mu);
                    if (TRACK) {
                        trk.after_nu = p.nu; trk.after_mu = p.mu; trk.after_energy = p.energy;
                        trk.line_emit_id = p.next_line_id - 1;
                        trk.interactions_count += 1 + trk.pop();
                        trk.radius = p.r; trk.nu = p.nu; trk.energy = p.energy; trk.shell_id = p.shell;
                        trk.interaction_type = IT_LINE;
                    }
                }
            } else {  // IT_ESCATTERING: thomson_scatter (interaction_events.py:184-217)
                if (TRACK) {
                    trk.before_mu = p.mu; trk.before_nu = p.nu; trk.before_energy = p.energy;
                    trk.line_absorb_id = -1; trk.line_emit_id = -1;
                }
                double vel = p.r / t;
                double old_dop = doppler_factor<FULL>(vel, p.mu);
                double comov_nu = p.nu * old_dop;
                double comov_energy = p.energy * old_dop;
                p.mu = 2.0 * rng.random(j) - 1.0;
                double inv_new = inverse_doppler_factor<FULL>(vel, p.mu);
                p.nu = comov_nu * inv_new;
                p.energy = comov_energy * inv_new;
                if (FULL) p.mu = aberration_cmf_to_lf(p.r, t, p.mu);
                if (TRACK) {
                    trk.after_mu = p.mu; trk.after_nu = p.nu; trk.after_energy = p.energy;
                    trk.interactions_count += 1 + trk.pop();
                    trk.radius = p.r; trk.nu = p.nu; trk.energy = p.energy; trk.shell_id = p.shell;
                    trk.interaction_type = IT_ESCATTERING;
                }
            }
        }
        if (err) {
            const long long i = chunk_first + pkt;
            if (j == 0) {
                atomicMin(&P.first_error[0], i);
                P.out_nu[i] = (double)err;
                P.out_e[i] = -99.0;
            }
            p.status = ST_EMITTED;
        } else if (p.status != ST_IN_PROCESS) {
            // set_packet_collection_output (modes/montecarlo_transport.py:70-90)
            const long long i = chunk_first + pkt;
            if (j == 0) {
                P.out_nu[i] = p.nu;
                P.out_e[i] = (p.status == ST_REABSORBED) ? -p.energy : p.energy;
                if (TRACK) {
                    P.li_radius[i] = trk.radius; P.li_nu[i] = trk.nu; P.li_energy[i] = trk.energy;
                    P.li_before_nu[i] = trk.before_nu; P.li_before_mu[i] = trk.before_mu; P.li_before_energy[i] = trk.before_energy;
                    P.li_after_nu[i] = trk.after_nu; P.li_after_mu[i] = trk.after_mu; P.li_after_energy[i] = trk.after_energy;
                    P.li_shell_id[i] = trk.shell_id; P.li_interaction_type[i] = trk.interaction_type;
                    P.li_line_absorb_id[i] = trk.line_absorb_id; P.li_line_emit_id[i] = trk.line_emit_id;
                    P.li_interactions_count[i] = trk.interactions_count;
                }
            }
        }
    }
    draws_total += (unsigned long long)rng.draws;
    __syncthreads();
    for (int s = threadIdx.x; s < P.n_shells; s += blockDim.x) {
        if (lds_J[s] != 0.0) atomic_add_f64(&P.J[s], lds_J[s]);
        if (lds_nubar[s] != 0.0) atomic_add_f64(&P.nubar[s], lds_nubar[s]);
    }
    if (j == 0) {
        atomicAdd(&P.counters[0], cn.visits);
        atomicAdd(&P.counters[1], cn.events);
        atomicAdd(&P.counters[2], cn.macro);
        atomicAdd(&P.counters[5], draws_total);
    }
}

}  // namespace mc
