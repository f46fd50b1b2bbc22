// propagate_group.hpp -- cooperative propagation kernel (variant 1): G = 16 or 8 lanes per packet.
//
// Why: the dominant cost of the lane-per-packet kernel is the 2 scattered fp64 atomics per line visit (measured
// ceiling on MI355X: ~24 G random fp64 atomics/s chip-wide, but ~170 G/s when 16 lanes hit 16 consecutive doubles,
// profiles/r01_microbench_and_ablation_lane_kernel.txt).  Here a G-lane group owns one packet and sweeps the sorted
// line list G lines at a time:
//   * nu_line[cur..cur+G) and tau[shell][cur..cur+G) are two coalesced loads (the next chunk is prefetched),
//   * every lane evaluates "its" line (distance, estimator energy),
//   * the running Sobolev optical depth is carried across lanes IN THE REFERENCE'S SERIAL ORDER (bit-exact),
//   * a ballot picks the first line at which the reference's loop would have stopped,
//   * lanes before it issue the j_blue / Edotlu atomics as contiguous groups.
// The packet's scalar event code (boundary distance, tau_event, move, scatter) is executed redundantly by the G
// lanes, so no cross-lane traffic is needed for it; the macro-atom block walk is cooperative again.  The packet's
// MT19937 state stays in global memory and is regenerated cooperatively G words at a time; raw seeded
// states are produced by a separate lane-per-packet kernel (the init_genrand recurrence is serial per packet).
// The last-interaction tracker lives in LDS (written by lane 0 on interactions only).
#pragma once
#include "mc_device.hpp"
#include "propagate_lane.hpp"  // xcc_id

namespace mc {

constexpr int PACKET_BATCH = 16;  // packets reserved per global atomic

// ---- seeding kernel: raw init_genrand state, [packet][624] contiguous, one packet per lane.
// Stores go through an LDS tile so that a 16-lane group writes 64 contiguous bytes of one packet's state.
__global__ void __launch_bounds__(256) seed_states_kernel(const uint32_t *__restrict__ seeds, uint32_t *__restrict__ states,
                                                          long long first, long long count, int stride)
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
                states[(size_t)gi * stride + base + (threadIdx.x & 15)] = tile[pk][threadIdx.x & 15];
        }
        __syncthreads();
    }
}

// ---- MT19937, one stream per packet.  The 624-word state stays in the packet's slot of the seeded-state buffer
// (global memory, touched by this group only) and is regenerated in place G words at a time with coalesced group
// loads/stores; the freshly regenerated block also stays in registers (one word per lane), so a draw is two
// cross-lane reads and no memory access.  No LDS is needed, which is what lets many groups share a CU.
template <int G>
struct GroupRng {
    uint32_t *st;    // this packet's 624-word state (global)
    uint32_t blk;    // lane j holds word (generated - G + j) of the stream (untempered)
    int consumed;    // stream words consumed since attach (group-uniform, even)
    int generated;   // stream words regenerated since attach (group-uniform, multiple of G, >= consumed)
    int cpos, gpos;  // consumed % 624, generated % 624
    int draws;

    __device__ __forceinline__ void attach(uint32_t *state) { st = state; consumed = generated = cpos = gpos = 0; blk = 0; }
    __device__ __forceinline__ void regenerate(int j)
    {   // next G words of the stream, in place at [gpos, gpos+G); 624 = 39*16 = 78*8.  Never call with
        // generated + G > consumed + 624 (it would overwrite unconsumed words).
        const int k = gpos + j;
        const int k1 = (k + 1 == MT_N) ? 0 : k + 1;
        const int km = (k + 397 >= MT_N) ? k + 397 - MT_N : k + 397;
        // make this wave's earlier stores to the state visible to these loads (same wave: waitcnt only)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t a = st[k], b = st[k1], c = st[km];
        uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        uint32_t v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        // word 623 needs the NEW word 0 (regenerated 38 steps ago); every other k+1 must be the OLD value, which
        // holds because all G loads above are issued before the G stores below (same wave, program order).
        st[k] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        blk = v;
        generated += G;
        gpos = (gpos + G == MT_N) ? 0 : gpos + G;
    }
    __device__ static __forceinline__ double to_double(uint32_t a, uint32_t b)
    {
        a ^= a >> 11; a ^= (a << 7) & 0x9d2c5680u; a ^= (a << 15) & 0xefc60000u; a ^= a >> 18;
        b ^= b >> 11; b ^= (b << 7) & 0x9d2c5680u; b ^= (b << 15) & 0xefc60000u; b ^= b >> 18;
        a >>= 5; b >>= 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ __forceinline__ void advance(int words)
    {
        consumed += words;
        cpos += words;
        if (cpos >= MT_N) cpos -= MT_N;
    }
    __device__ __forceinline__ double random(int j)
    {
        if (consumed == generated) regenerate(j);
        uint32_t a, b;
        if (generated - consumed <= G) {  // the pair is in the register-resident block
            const int w = cpos & (G - 1);
            a = (uint32_t)__shfl((int)blk, w, G);
            b = (uint32_t)__shfl((int)blk, w + 1, G);
        } else {  // regenerated ahead (v-packet volleys peek): read it back from the state
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            a = st[cpos];
            b = st[cpos + 1];
        }
        advance(2);
        ++draws;
        return to_double(a, b);
    }
    // ---- look-ahead used by the v-packet volleys: the n-th double after the current position, per lane
    __device__ __forceinline__ void ensure_ahead(int words, int j)
    {   // group-cooperative: make words [consumed, consumed + words) available (words <= 624 - 2 G)
        while (generated - consumed < words) regenerate(j);
    }
    __device__ __forceinline__ double peek(int n) const
    {
        int w = cpos + 2 * n;
        if (w >= MT_N) w -= MT_N;
        return to_double(st[w], st[w + 1]);
    }
};

// last-interaction tracker in LDS (packets/trackers/tracker_last_interaction.py:8-254), one per group
struct LdsTracker {
    double radius, nu, energy, before_nu, before_mu, before_energy, after_nu, after_mu, after_energy;
    int shell_id, interaction_type, line_absorb_id, line_emit_id, interactions_count, boundary_buffer;
    int pad[2];
};

template <int G> __device__ __forceinline__ double gbcast(double v, int src) { return __shfl(v, src, G); }
template <int G> __device__ __forceinline__ int gbcast(int v, int src) { return __shfl(v, src, G); }

// lane j <- value of lane j-1 of its 16-lane DPP row; lane 0 of the row keeps `first`
__device__ __forceinline__ double dpp_row_shr1(double first, double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x111, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the same with +0.0 shifted into lane 0 of the row (bound_ctrl): no fill operand to set up
__device__ __forceinline__ double dpp_row_shr1_zero(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x111, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ((carry + v0) + v1) + ... + vj for lane j: the reference's sequential accumulation order, evaluated exactly.
// acc <- shr1(acc) + v repeated G-1 times: after step s lanes 0..s hold their final value and recomputing a final
// lane from its (final) left neighbour reproduces the same value, so no per-step masking is needed.
// lane j <- value of lane j-1 of its 16-lane DPP row, AND-ed with a per-lane bit mask; +0.0 is shifted into lane 0 of the
// row.  Written so that the DPP move folds into the v_and (one instruction per 32-bit half).
__device__ __forceinline__ double dpp_row_shr1_and(double v, int mask)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true) & mask;
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true) & mask;
    return __hiloint2double(hi, lo);
}

template <int G>
__device__ __forceinline__ double serial_prefix(double carry, double v, int j)
{
    const double head = carry + v;  // value of the group's first lane
    double acc = head;
    // lane 0 adds its head to +0.0 (exact), every other lane adds its own v to its left neighbour's running value
    const double addend = (j == 0) ? head : v;
    if (G == 16) {
#pragma unroll
        for (int s = 1; s < G; ++s) acc = dpp_row_shr1_zero(acc) + addend;
    } else {
        // a smaller group shares its DPP row with neighbour groups: what arrives from the neighbour is masked to +0.0
        int keep = (j == 0) ? 0 : -1;
        asm volatile("" : "+v"(keep));  // keep it a plain bit mask: v_and_b32 can take the DPP operand, v_cndmask cannot
#pragma unroll
        for (int s = 1; s < G; ++s) acc = dpp_row_shr1_and(acc, keep) + addend;
    }
    return acc;
}

// value of the left neighbour in the group (lane 0 gets `first`)
template <int G>
__device__ __forceinline__ double group_shr1(double first, double v, int j)
{
    double r = dpp_row_shr1(first, v);
    if (G < 16) r = (j == 0) ? first : r;
    return r;
}

struct GroupCounters { unsigned long long visits = 0; unsigned events = 0, macro = 0; };

// ---- one cooperative trace_packet (modes/homologous_rad_packet_transport.py:30-174)
// All arguments / results are group-uniform except the lane index j.  Returns 0 or a negative error code.
template <bool FULL, int G, bool FAST>
__device__ __forceinline__ int sweep_lines(const GroupArgs &P, Packet &p, const int j, const double chi_cont,
                                           const double tau_event, const double comov_nu, const double d_boundary,
                                           int line, bool in_range, double nu_line, double tau_line,
                                           double *__restrict__ jb, double *__restrict__ ed, double &distance, int &type,
                                           GroupCounters &cn);

template <bool FULL, int G>
__device__ __forceinline__ int trace_packet_group(const GroupArgs &P, Packet &p, GroupRng<G> &rng, const int j,
                                                  const double chi_cont, const double r_inner, const double r_outer,
                                                  const double dop, double *__restrict__ jb, double *__restrict__ ed,
                                                  double &distance, int &type, int &delta_shell, GroupCounters &cn)
{
    const int L = P.n_lines;
    const int start = p.next_line_id;
    // software pipeline: chunk c+1 is loaded while chunk c is evaluated; the first loads fly during the prologue.
    // (Aligning chunks to G-line memory segments was measured: fewer atomic/load requests but more chunk steps, net loss.)
    int line = start + j;
    bool in_range = line < L;
    double nu_line = in_range ? P.nu_line[(unsigned)line] : 0.0;
    double tau_line = in_range ? P.tau_t[(unsigned)p.shell * (unsigned)L + (unsigned)line] : 0.0;

    double d_boundary;
    distance_boundary(p.r, p.mu, r_inner, r_outer, d_boundary, delta_shell);
    const double tau_event = -mcm::log(rng.random(j));
    const double comov_nu = p.nu * dop;
    cn.events++;
    // the 3-instruction exact division needs operands away from the exponent limits (always true for physical input)
    const bool fast = mid_range(p.nu) && mid_range(chi_cont) && mid_range(tau_event) && mid_range(p.energy) &&
                      mid_range(p.r) && mid_range(comov_nu) && mid_range(P.t_exp) && !(P.debug_flags & 4);
    if (fast) return sweep_lines<FULL, G, true>(P, p, j, chi_cont, tau_event, comov_nu, d_boundary, line, in_range, nu_line,
                                               tau_line, jb, ed, distance, type, cn);
    return sweep_lines<FULL, G, false>(P, p, j, chi_cont, tau_event, comov_nu, d_boundary, line, in_range, nu_line, tau_line,
                                       jb, ed, distance, type, cn);
}

// The line sweep of trace_packet (lines 100-172 of homologous_rad_packet_transport.py), G lines per step.
template <bool FULL, int G, bool FAST>
__device__ __forceinline__ int sweep_lines(const GroupArgs &P, Packet &p, const int j, const double chi_cont,
                                           const double tau_event, const double comov_nu, const double d_boundary,
                                           int line, bool in_range, double nu_line, double tau_line,
                                           double *__restrict__ jb, double *__restrict__ ed, double &distance, int &type,
                                           GroupCounters &cn)
{
    const int L = P.n_lines;
    const double t = P.t_exp;
    const int start = p.next_line_id;
    const unsigned row = (unsigned)p.shell * (unsigned)L;  // 32-bit element offset of this shell's row (S*L < 2^28)
    const double *__restrict__ tau_t = P.tau_t;
    const double mur = p.mu * p.r;
    const double tc = P.tc, rcp_tc = P.rcp_tc;
    // reciprocals for the exact 3-instruction divisions of the sweep (the divisors are fixed during one trace)
    const double rcp_nu = 1.0 / p.nu;
    const double rcp_chi = 1.0 / chi_cont;
    const int last = L - 1;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : 0xffull;
    double tau_carry = 0.0;                      // tau_trace_line_combined before the first line of this chunk
    double d_cont_carry = exact_div<FAST>(tau_event, chi_cont, rcp_chi);  // distance_continuous in force at the first line of this chunk
    // The estimator atomics of a chunk are issued one chunk late, right after the wait for the next chunk's data: vector
    // memory waits are in issue order (and loads mixed with atomics force vmcnt(0)), so atomics issued just before such a
    // wait would put their full memory-side latency on the critical path; issued just after it they overlap a whole
    // chunk of arithmetic.
    bool pend_valid = false;
    unsigned pend_idx = 0;
    double pend_energy = 0.0;
    auto flush_pending = [&]() {
        if (pend_valid && !(P.debug_flags & 1)) {
            atomic_add_f64(&jb[pend_idx], exact_div<FAST>(pend_energy, p.nu, rcp_nu));
            atomic_add_f64(&ed[pend_idx], pend_energy);
        }
        pend_valid = false;
    };

    for (int cur0 = start; cur0 < L; cur0 += G) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this chunk's prefetched data (and all older atomics) have landed
        flush_pending();
        // prefetch the next chunk
        const int nline = cur0 + G + j;
        const bool nin = nline < L;
        const double nu_next = nin ? P.nu_line[(unsigned)nline] : 0.0;
        const double tau_next = nin ? tau_t[row + (unsigned)nline] : 0.0;

        const double tau_incl = serial_prefix<G>(tau_carry, tau_line, j);
        const double tau_prev = group_shr1<G>(tau_carry, tau_incl, j);  // tau_trace_line_combined before this lane's line
        const double d_cont = (j == 0) ? d_cont_carry : exact_div<FAST>(tau_event - tau_prev, chi_cont, rcp_chi);
        // calculate_distance_line (calculate_distances.py:66-112), select form
        const bool is_last = line == last;
        const double nu_diff = comov_nu - nu_line;
        const double q = exact_div<FAST>(nu_diff, p.nu, rcp_nu);
        const bool close = fabs(q) < CLOSE_LINE_THRESHOLD;
        const bool err = in_range && !is_last && !close && !(nu_diff >= 0);
        double d_far;
        if (FULL) d_far = distance_line_full_relativity(nu_line, p.nu, t, p.r, p.mu);
        else d_far = q * C_LIGHT * t;
        const double d_trace = is_last ? MISS_DISTANCE : (close ? 0.0 : d_far);
        const double tau_combined = tau_incl + chi_cont * d_trace;
        double dmin = d_trace;  // Python min(d_trace, d_boundary, d_cont)
        if (d_boundary < dmin) dmin = d_boundary;
        if (d_cont < dmin) dmin = d_cont;
        const bool ok = in_range && !err;
        const bool stop_b = ok && d_trace != 0 && dmin == d_boundary;
        const bool stop_e = ok && d_trace != 0 && !stop_b && dmin == d_cont;
        const bool stop_l = ok && !stop_b && !stop_e && tau_combined > tau_event && !P.disable_line_scattering;
        const bool stop = stop_b || stop_e || stop_l || (in_range && err);
        const unsigned stop_mask = (unsigned)((__ballot(stop) >> gshift) & GMASK);
        const int first = stop_mask ? __builtin_ctz(stop_mask) : G;  // group-uniform
        const int code = stop_b ? 1 : (stop_e ? 2 : (stop_l ? 3 : ((in_range && err) ? 4 : 0)));
        const int first_code = gbcast<G>(code, first & (G - 1));
        // lines before the stopping one are passed (estimators updated); a LINE stop updates its own line too
        const bool visited = in_range && (j < first || (j == first && first_code == 3));
        pend_valid = visited;
        pend_idx = row + (unsigned)line;
        if (!FULL) pend_energy = p.energy * (1.0 - exact_div<FAST>(d_trace + mur, tc, rcp_tc));
        else pend_energy = p.energy;
        if (first < G) {
            flush_pending();
            cn.visits += (unsigned long long)(first + 1);
            if (first_code == 4) return ERR_MONTECARLO;
            p.next_line_id = cur0 + first;
            if (first_code == 1) { type = IT_BOUNDARY; distance = d_boundary; }
            else if (first_code == 2) { type = IT_ESCATTERING; distance = gbcast<G>(d_cont, first); }
            else { type = IT_LINE; distance = gbcast<G>(d_trace, first); }
            return 0;
        }
        // whole chunk passed: carry the running optical depth and the continuum distance into the next chunk
        const int n_in = min(G, L - cur0);
        cn.visits += (unsigned long long)n_in;
        tau_carry = gbcast<G>(tau_incl, n_in - 1);
        d_cont_carry = exact_div<FAST>(tau_event - tau_carry, chi_cont, rcp_chi);
        line = nline; in_range = nin; nu_line = nu_next; tau_line = tau_next;
    }
    flush_pending();
    // for-else (lines 157-172): the line list is exhausted; next_line_id is left untouched
    if (d_cont_carry < d_boundary) { distance = d_cont_carry; type = IT_ESCATTERING; }
    else { distance = d_boundary; type = IT_BOUNDARY; }
    return 0;
}

// macro_atom_interaction (macro_atom.py:52-104): the transition block of the activated level is loaded G
// probabilities at a time (together with the packed transition records), accumulated in the reference's serial
// order, and a ballot finds the selected row.  The record carries the destination level's block bounds, so an
// internal jump costs ONE dependent memory round trip instead of three (edge, probability, destination).
template <int G>
__device__ __forceinline__ int macro_atom_group(const GroupArgs &P, GroupRng<G> &rng, const int j, int b0, int b1, int shell,
                                                int &out_line, GroupCounters &cn)
{
    const unsigned row = (unsigned)shell * (unsigned)P.n_trans;
    const double *__restrict__ prob_t = P.prob_t;
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : 0xffull;
    for (;;) {
        const double event = rng.random(j);
        double carry = 0.0;
        int ttype = 0;
        bool found = false;
        for (int base = b0; base < b1; base += G) {
            const int k = base + j;
            const bool in = k < b1;
            const double pr = in ? prob_t[row + (unsigned)k] : 0.0;
            int4 rec = make_int4(0, 0, 0, 0);
            if (in) rec = P.trans_rec[(unsigned)k];
            const double acc = serial_prefix<G>(carry, pr, j);
            const unsigned hit = (unsigned)((__ballot(in && acc > event) >> gshift) & GMASK);
            if (hit) {
                const int f = __builtin_ctz(hit);
                cn.macro += (unsigned)(f + 1);
                out_line = gbcast<G>(rec.x, f);
                ttype = gbcast<G>(rec.y, f);
                b0 = gbcast<G>(rec.z, f);
                b1 = gbcast<G>(rec.w, f);
                found = true;
                break;
            }
            const int n_in = min(G, b1 - base);
            cn.macro += (unsigned)n_in;
            carry = gbcast<G>(acc, n_in - 1);
        }
        if (!found) return ERR_MACRO_ATOM;
        if (ttype < 0) return ttype == -1 ? 0 : ERR_UNSUPPORTED;
    }
}

// ---- v-packets (packets/virtual_packet.py:82-386).  Inside a volley every LANE of the group traces one v-packet (the
// per-v-packet work is a serial sum over lines, which the reference's accumulation order does not let us split), i.e.
// up to G v-packets of one volley advance together.  The n_v mu-draws and the Russian-roulette draws all come from the
// parent packet's stream in sequence; lane i therefore reads its draws at the stream position it would have IF no
// earlier v-packet of the round played roulette.  After the round the first lane that did consume a roulette draw
// invalidates the lanes after it, which are simply re-traced in the next round from the corrected stream position.
struct VpDraws {  // draws of ONE v-packet: position `first` (in doubles after the group's current stream position)
    int first, used, limit;
    bool overflow;
};

constexpr int VP_SUM_BATCH = 16;  // optical depths a lane requests per round trip of a v-packet's per-shell sum

// The four-line window around the frequency-bucket guess did not pin the first line after `start` whose resonance lies at or beyond the
// shell boundary (`stops`: monotone along the sorted list): it lies before the window (before_window: stops(w0) held and w0 > start + 1) or
// beyond it.  The reference walks there line by line (virtual_packet.py:132-150); here FOUR lines per dependent round trip, backwards or
// forwards -- in a wave every lane waits for the longest of these walks, and with three lines per bucket 22 % of the crossings took
// one (profiles/r05_bucket_index.txt).  Returns the stopping line, or L when no line stops (a NaN boundary distance: the reference then sums
// every line).
template <typename Stops /* bool(int k, double nu_k) */>
__device__ __forceinline__ int vp_walk_to_stop(const MC_G double *__restrict__ nu_line_g, int L, int start, int w0, bool before_window, Stops &&stops)
{
    typedef double nu2 __attribute__((ext_vector_type(2), aligned(8)));
    auto first_of_four = [&](int b) -> int {  // index 0..3 of the first stopping line among b .. b + 3, 4: none
        const nu2 a = *reinterpret_cast<const MC_G nu2 *>(nu_line_g + (unsigned)b), c = *reinterpret_cast<const MC_G nu2 *>(nu_line_g + (unsigned)b + 2);
        const double w[4] = {a.x, a.y, c.x, c.y};
        int f = 4;
#pragma unroll
        for (int i = 3; i >= 0; --i) if (stops(min(b + i, L - 1), w[i])) f = i;
        return f;
    };
    if (before_window) {
        const int lo = start + 1;
        int hi = w0;  // stops(hi) holds
        for (;;) {
            const int b = max(lo, hi - 4);
            const int f = first_of_four(b);
            if (f > 0 || b == lo) return min(b + f, hi);  // (f == 4: none of b .. hi - 1 stops)
            hi = b;
        }
    }
    for (int b = w0 + 4;; b += 4) {  // (the window w0 .. w0 + 3 did not stop)
        if (b > L - 1) return L;     // (not even the last line: only with a NaN boundary distance)
        const int f = first_of_four(b);
        if (f < 4) return min(b + f, L - 1);
    }
}

template <bool FULL, int G>
__device__ __forceinline__ int vp_trace(const GroupArgs &P, const GroupRng<G> &rng, VpDraws &dr, double r, double mu, double nu,
                                        double &energy, int shell, int next_line, double &tau_out, unsigned &vvisits, const double *geo)
{
    const int L = P.n_lines;
    const double t = P.t_exp;
    double tau = 0.0;
    int status = ST_IN_PROCESS;
    for (;;) {
        // trace_vpacket_within_shell (:82-175).  The reference walks the lines one by one until the first one whose
        // resonance distance is >= the boundary distance and adds up their tau in that order.  The stopping predicate is
        // monotone along the (descending) line list, so the stopping index is located through the frequency-bucket
        // index and then pinned with the reference's own predicate; the tau sum keeps the reference's order.
        // (a crossing is a chain of dependent round trips: the line at `start` is requested together with the geometry, and the
        // frequency bucket of the boundary as soon as the boundary distance is there -- not behind the test of the line at `start`)
        const int start = next_line;
        const double nl_start = P.nu_line[(unsigned)min(start, L - 1)];
        double d_boundary;
        int delta;
        distance_boundary(r, mu, geo[shell], geo[P.n_shells + shell], d_boundary, delta);
        const double chi_e = geo[2 * P.n_shells + shell] * P.sigma_thomson;
        const double velocity = r / t;
        const double dop = doppler_factor<FULL>(velocity, mu);
        const double comov_nu = nu * dop;
        double chi_cont = chi_e;
        if (FULL) chi_cont *= dop;
        double tau_shell = chi_cont * d_boundary;
        const unsigned row = (unsigned)shell * (unsigned)L;
        // approximate stopping frequency: nu_line ~ comov_nu - d_boundary nu / (c t)
        const double nu_thr = comov_nu - d_boundary * P.rcp_tc * nu;
        long long kk = (long long)((unsigned long long)__double_as_longlong(nu_thr > 0.0 ? nu_thr : 0.0) >> P.bucket_shift) - P.bucket_kmin;
        kk = kk < 0 ? 0 : (kk >= P.bucket_n ? P.bucket_n - 1 : kk);
        const int bucket_e = P.bucket_first[kk];
        if (start < L) {
            double d_line;
            // the reference evaluates line `start` first and raises there if it lies blueward of the packet
            if (!distance_line<FULL>(nu, r, mu, comov_nu, start == L - 1, nl_start, t, d_line)) return ERR_MONTECARLO;
            int e = start;
            if (!(d_boundary <= d_line)) {
                e = max(bucket_e, start + 1);
                if (e > L - 1) e = L - 1;
                // (lines after `start` cannot raise: the list is sorted, their nu_diff is larger than that of `start`)
                auto stops_at_nu = [&](int k, double nl) -> bool {
                    double d;
                    (void)distance_line<FULL>(nu, r, mu, comov_nu, k == L - 1, nl, t, d);
                    return d_boundary <= d;
                };
                auto stops_at = [&](int k) -> bool { return stops_at_nu(k, P.nu_line[(unsigned)k]); };
                // a window of four lines around the bucket guess in ONE round trip (the walk below -- one dependent load per
                // line -- only if the guess was further off); two 16-byte loads: the list ends in slack, and an index clamped
                // to the last line does not look at its frequency
                const int w0 = max(e - 1, start + 1);
                typedef double nu2 __attribute__((ext_vector_type(2), aligned(8)));
                const nu2 wa = *reinterpret_cast<const nu2 *>(P.nu_line + (unsigned)w0), wb = *reinterpret_cast<const nu2 *>(P.nu_line + (unsigned)w0 + 2);
                const double wn[4] = {wa.x, wa.y, wb.x, wb.y};
                bool sw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) sw[i] = stops_at_nu(min(w0 + i, L - 1), wn[i]);
                bool resolved = false, stops = false;
                if (sw[0]) {
                    if (w0 == start + 1) { e = w0; stops = true; resolved = true; }
                    else e = w0;  // the first stopping line lies before the window
                } else if (sw[1]) { e = min(w0 + 1, L - 1); stops = true; resolved = true; }
                else if (sw[2]) { e = min(w0 + 2, L - 1); stops = true; resolved = true; }
                else if (sw[3]) { e = min(w0 + 3, L - 1); stops = true; resolved = true; }
                else e = min(w0 + 3, L - 1);  // beyond the window
                if (!resolved) {  // the bucket guess was further off: the reference's walk, four lines per round trip
                    e = vp_walk_to_stop(glob(P.nu_line), L, start, w0, sw[0], stops_at_nu);
                    stops = e < L;
                }
                if (!stops) e = L;  // (only with a NaN/huge boundary distance: the reference then sums every line)
            }
            // serial-order sum of tau over [start, e): the loads are independent of the adds, so VP_SUM_BATCH of them are in flight
            // per round trip (a shell crossing of the 5e5-line list passes ~40 lines: with four at a time the sum alone was a chain
            // of ten dependent round trips)
            // (two optical depths per load instruction: the address unit's time goes by lanes x instructions, and these per-lane
            // runs of a row are what keeps it 65 % busy on the 100-shell shape; rows are 8-byte aligned, the table ends in slack)
            typedef double tau2 __attribute__((ext_vector_type(2), aligned(8)));
            const double *__restrict__ trow = P.tau_t + row;
            int k = start;
            const int e_sum = min(e, L);
            for (; k + VP_SUM_BATCH <= e_sum; k += VP_SUM_BATCH) {
                tau2 tb[VP_SUM_BATCH / 2];
#pragma unroll
                for (int q = 0; q < VP_SUM_BATCH / 2; ++q) tb[q] = *reinterpret_cast<const tau2 *>(trow + (unsigned)(k + 2 * q));
#pragma unroll
                for (int q = 0; q < VP_SUM_BATCH / 2; ++q) { tau_shell += tb[q].x; tau_shell += tb[q].y; }
            }
            if (k < e_sum) {  // the rest (< VP_SUM_BATCH lines), again in one round trip; +0.0 where there is no line
                tau2 tb[VP_SUM_BATCH / 2];
#pragma unroll
                for (int q = 0; q < VP_SUM_BATCH / 2; ++q) {
                    const tau2 z = {0.0, 0.0};
                    tb[q] = (k + 2 * q < e_sum) ? *reinterpret_cast<const tau2 *>(trow + (unsigned)(k + 2 * q)) : z;
                    if (!(k + 2 * q + 1 < e_sum)) tb[q].y = 0.0;
                }
#pragma unroll
                for (int q = 0; q < VP_SUM_BATCH / 2; ++q) { tau_shell += tb[q].x; tau_shell += tb[q].y; }
            }
            vvisits += (unsigned)((e < L) ? (e - start + 1) : (L - start));
            next_line = e;
        }
        // trace_vpacket (:179-244)
        tau += tau_shell;
        cross_shell(shell, status, delta, P.n_shells);
        if (tau > P.tau_russian) {
            double ev = 0.0;
            if (dr.used < dr.limit) ev = rng.peek(dr.first + dr.used); else dr.overflow = true;
            dr.used++;
            if (ev > P.survival_probability) {
                energy = 0.0;
                status = ST_EMITTED;
            } else {
                energy = energy / P.survival_probability * mcm::exp(-tau);
                tau = 0.0;
            }
        }
        const double new_r = sqrt(r * r + d_boundary * d_boundary + 2.0 * r * d_boundary * mu);
        mu = (mu * r + d_boundary) / new_r;
        r = new_r;
        if (status == ST_EMITTED) break;
    }
    tau_out = tau;
    return 0;
}

// The same trace with the optical depths of a shell crossing taken from the prefix sums of the row (tau_prefix.hpp): decides the
// Russian roulette -- and with the default survival probability 0 the whole v-packet -- without reading the lines.
// Returns 1: the v-packet is dropped (energy 0; dr.used and vvisits as the line-by-line trace would leave them);
//         0: not decided here (it leaves the grid alive, or comes within the error margin of the threshold, or drew exactly 0.0):
//            the caller traces it line by line from the start; < 0: error (the reference raises).
// A v-packet never changes direction, so its whole path -- radii, angles, boundary distances, comoving frequencies at the
// boundaries -- is arithmetic on the geometry in LDS; only the line indices need memory.  One round trip per shell crossing: while
// the four-line window around the frequency-bucket guess of crossing k is in flight (frequencies, the prefix sums of this
// shell's row and of the next shell's row at the same indices: the stopping line of k is the start line of k + 1), so is the
// bucket look-up of crossing k + 1, whose geometry is computed ahead.
template <bool FULL, int G>
__device__ __forceinline__ int vp_screen(const GroupArgs &P, const GroupRng<G> &rng, VpDraws &dr, double r, double mu, double nu,
                                         int shell, int next_line, unsigned &vvisits, const double *geo)
{
    const int L = P.n_lines, S = P.n_shells;
    const double t = P.t_exp;
    // (the kernel is half bound by its VALU instructions at ~12 live lanes per wave: the divisions by t and by nu -- constant along
    // a v-packet -- take the exact three-instruction form of mc_device.hpp where their operands allow it)
    const double rcp_t = 1.0 / t, rcp_nu = 1.0 / nu;
    const bool fast_t = mid_range(t), fast_nu = mid_range(nu);
    typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
    struct Crossing { double d_boundary, comov_nu, tau_cont; int delta, bucket_e; };
    auto geometry = [&](double rr, double mm, int sh, Crossing &c) {
        distance_boundary(rr, mm, geo[sh], geo[S + sh], c.d_boundary, c.delta);
        const double dop = doppler_factor<FULL>((fast_t && mid_range(rr)) ? exact_div<true>(rr, t, rcp_t) : rr / t, mm);
        c.comov_nu = nu * dop;
        double chi_cont = geo[2 * S + sh] * P.sigma_thomson;
        if (FULL) chi_cont *= dop;
        c.tau_cont = chi_cont * c.d_boundary;
        const double nu_thr = c.comov_nu - c.d_boundary * P.rcp_tc * nu;
        long long kk = (long long)((unsigned long long)__double_as_longlong(nu_thr > 0.0 ? nu_thr : 0.0) >> P.bucket_shift) - P.bucket_kmin;
        kk = kk < 0 ? 0 : (kk >= P.bucket_n ? P.bucket_n - 1 : kk);
        c.bucket_e = P.bucket_first[kk];
    };
    double tau = 0.0, margin = 0.0;
    unsigned visits = 0;
    Crossing cur;
    geometry(r, mu, shell, cur);
    double nl_start = P.nu_line[(unsigned)min(next_line, L - 1)];
    double p_start = P.tau_pfx[(size_t)shell * (size_t)(L + 1) + (unsigned)min(next_line, L)];
    for (;;) {
        const int start = next_line;
        // where the v-packet is after this crossing, and the geometry (and bucket look-up) of the next one
        const double new_r = sqrt(r * r + cur.d_boundary * cur.d_boundary + 2.0 * r * cur.d_boundary * mu);
        const double new_mu = (mu * r + cur.d_boundary) / new_r;
        int status = ST_IN_PROCESS, nshell = shell;
        cross_shell(nshell, status, cur.delta, S);
        Crossing nxt = cur;
        const bool leaves = status == ST_EMITTED;
        if (!leaves) geometry(new_r, new_mu, nshell, nxt);
        const double *__restrict__ prow = P.tau_pfx + (size_t)shell * (size_t)(L + 1);
        const double *__restrict__ nrow = P.tau_pfx + (size_t)(leaves ? shell : nshell) * (size_t)(L + 1);
        double seg = 0.0, p_next = nrow[(unsigned)min(start, L)], nl_next = nl_start;
        int n_sum = 0;
        if (start < L) {
            double d_line;
            if (!distance_line<FULL>(nu, r, mu, cur.comov_nu, start == L - 1, nl_start, t, d_line)) return ERR_MONTECARLO;
            int e = start;
            if (!(cur.d_boundary <= d_line)) {
                e = max(cur.bucket_e, start + 1);
                if (e > L - 1) e = L - 1;
                // calculate_distance_line (calculate_distances.py:66-112) of a line after `start` (those cannot raise: the list is sorted)
                auto stops_at_nu = [&](int k, double nl) -> bool {
                    double d;
                    if (FULL) (void)distance_line<FULL>(nu, r, mu, cur.comov_nu, k == L - 1, nl, t, d);
                    else {
                        const double nu_diff = cur.comov_nu - nl;
                        const double q = (fast_nu && mid_range(nu_diff)) ? exact_div<true>(nu_diff, nu, rcp_nu) : nu_diff / nu;
                        d = (k == L - 1) ? MISS_DISTANCE : ((fabs(q) < CLOSE_LINE_THRESHOLD) ? 0.0 : q * C_LIGHT * t);
                    }
                    return cur.d_boundary <= d;
                };
                auto stops_at = [&](int k) -> bool { return stops_at_nu(k, P.nu_line[(unsigned)k]); };
                const int w0 = max(e - 1, start + 1);
                const dbl2 wa = *reinterpret_cast<const dbl2 *>(P.nu_line + (unsigned)w0), wb = *reinterpret_cast<const dbl2 *>(P.nu_line + (unsigned)w0 + 2);
                const dbl2 pa = *reinterpret_cast<const dbl2 *>(prow + (unsigned)w0), pb = *reinterpret_cast<const dbl2 *>(prow + (unsigned)w0 + 2);
                const dbl2 qa = *reinterpret_cast<const dbl2 *>(nrow + (unsigned)w0), qb = *reinterpret_cast<const dbl2 *>(nrow + (unsigned)w0 + 2);
                const double wn[4] = {wa.x, wa.y, wb.x, wb.y}, wp[4] = {pa.x, pa.y, pb.x, pb.y}, wq[4] = {qa.x, qa.y, qb.x, qb.y};
                bool sw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) sw[i] = stops_at_nu(min(w0 + i, L - 1), wn[i]);
                int hit = -1;  // window slot of the stopping line, when the window pins it
                if (sw[0]) { if (w0 == start + 1) hit = 0; }
                else if (sw[1]) hit = 1;
                else if (sw[2]) hit = 2;
                else if (sw[3]) hit = 3;
                if (hit >= 0 && w0 + hit <= L - 1) {
                    e = w0 + hit;
                    double pe = wp[0], qe = wq[0], ne = wn[0];
#pragma unroll
                    for (int i = 1; i < 4; ++i) if (hit == i) { pe = wp[i]; qe = wq[i]; ne = wn[i]; }
                    seg = pe - p_start; p_next = qe; nl_next = ne;
                } else {  // the bucket guess was further off (or the window ran past the list): the reference's walk, four lines per round trip
                    if (hit >= 0) e = L - 1;  // (the window ran past the list: its last line stops)
                    else e = vp_walk_to_stop(glob(P.nu_line), L, start, w0, sw[0], stops_at_nu);
                    seg = prow[(unsigned)min(e, L)] - p_start;
                    p_next = nrow[(unsigned)min(e, L)];
                    nl_next = P.nu_line[(unsigned)min(e, L - 1)];
                }
            }
            n_sum = min(e, L) - start;
            visits += (unsigned)((e < L) ? (e - start + 1) : (L - start));
            next_line = e;
        }
        const double tau_shell = cur.tau_cont + seg;
        tau += tau_shell;
        margin += 2.3e-16 * (geo[3 * S + shell] + (double)(n_sum + 4) * tau_shell + 2.0 * tau);
        if (tau - 2.0 * margin > P.tau_russian) {  // the reference's `tau_trace_combined > tau_russian` is certainly true
            double ev = 0.0;
            if (dr.used < dr.limit) ev = rng.peek(dr.first + dr.used); else dr.overflow = true;
            dr.used++;
            if (!(ev > P.survival_probability)) { dr.used = 0; dr.overflow = false; return 0; }  // (a draw of exactly 0.0)
            vvisits += visits;
            return 1;
        }
        if (!(tau + 2.0 * margin < P.tau_russian)) return 0;  // too close to call
        if (leaves) return 0;                                 // leaves the grid alive: its energy needs the reference's own sum
        r = new_r; mu = new_mu; shell = nshell;
        cur = nxt; p_start = p_next; nl_start = nl_next;
    }
}

template <bool FULL, int G>
__device__ __forceinline__ int volley_group(const GroupArgs &P, const Packet &p, GroupRng<G> &rng, const int j, long long packet_index,
                                            int &vseq, unsigned &pred_bits, unsigned &vvisits, unsigned &vcount, unsigned long long &vtraced,
                                            const double *geo)
{   // trace_vpacket_volley (:248-386)
    if (p.nu < P.spawn_start || p.nu > P.spawn_end) return 0;
    const int n_v = (int)P.n_vpackets;
    if (n_v == 0) return 0;
    const double t = P.t_exp;
    double mu_min, beta_inner = 0.0;
    bool on_inner;
    const double r_in0 = P.r_inner[0];
    if (p.r > r_in0) {
        double r_inner_over_r = r_in0 / p.r;
        mu_min = -sqrt(1 - r_inner_over_r * r_inner_over_r);
        on_inner = false;
        if (FULL) mu_min = aberration_lf_to_cmf(p.r, t, mu_min);
    } else {
        on_inner = true;
        mu_min = 0.0;
        if (FULL) {
            const double inv_c = 1 / C_LIGHT;
            double inv_t = 1 / t;
            beta_inner = r_in0 * inv_t * inv_c;
        }
    }
    const double mu_bin = (1.0 - mu_min) / (double)n_v;
    const double r_velocity = p.r / t;
    const double r_dop = doppler_factor<FULL>(r_velocity, p.mu);
    const int gshift = (threadIdx.x & 63) & ~(G - 1);
    constexpr unsigned long long GMASK = (G == 16) ? 0xffffull : 0xffull;
    constexpr int ROULETTE_SLACK = 8;  // spare draws available to a round beyond one mu + one roulette draw per v-packet
    const DeviceProblem *C = P.cold;

    int done = 0;    // v-packets of this volley finalised so far (group-uniform)
    int err_out = 0;
    while (done < n_v) {
        const int n_round = min(G, n_v - done);
        rng.ensure_ahead(2 * (2 * n_round + ROULETTE_SLACK), j);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the peeks below read what regenerate() stored
        const int i = done + j;  // this lane's v-packet
        const bool active = j < n_round;
        // Predicted roulette draws consumed by the v-packets of this round that precede lane j.  v-packet i of every volley
        // is launched into the same mu bin, so whether it played roulette the last time it was traced (previous volley of
        // this packet, or a discarded speculative trace of this volley) predicts whether it will now: bit i of pred_bits.
        const unsigned round_bits = (pred_bits >> done) & ((1u << j) - 1u);  // v-packets done .. i-1
        const int extra_before = __popc(round_bits);
        const int pred_self = (int)((pred_bits >> i) & 1u);
        int err = 0;
        VpDraws dr;
        dr.first = j + extra_before; dr.used = 0; dr.overflow = false;
        dr.limit = (2 * n_round + ROULETTE_SLACK) - dr.first - 1;
        double v_nu = 0.0, v_energy = 0.0, v_mu0 = 0.0;
        unsigned my_visits = 0;
        if (active) {
            const double xi = rng.peek(dr.first);
            dr.first += 1;  // roulette draws follow the mu draw
            double v_mu = mu_min + (double)i * mu_bin + xi * mu_bin;
            double weight;
            if (on_inner) {
                if (!FULL) weight = 2 * v_mu / (double)n_v;
                else weight = 2 * (v_mu + beta_inner) / (2 * beta_inner + 1) / (double)n_v;
            } else
                weight = (1 - mu_min) / (double)(2 * n_v);
            if (FULL) v_mu = aberration_cmf_to_lf(p.r, t, v_mu);
            v_mu0 = v_mu;  // the log records the (aberrated) launch direction (virtual_packet.py:337-340,375)
            const double v_dop = doppler_factor<FULL>(r_velocity, v_mu);
            const double ratio = r_dop / v_dop;
            v_nu = p.nu * ratio;
            v_energy = p.energy * weight * ratio;
            int screened = 0;
            // (only v-packets expected to be dropped: bit i of the roulette predictor -- a v-packet that leaves the grid alive
            // needs the line-by-line trace anyway, and the screening would be a second trace on top of it)
            if (P.tau_pfx && pred_self) screened = vp_screen<FULL, G>(P, rng, dr, p.r, v_mu, v_nu, p.shell, p.next_line_id, my_visits, geo);
            if (screened > 0) {
                v_energy = 0.0;  // dropped by the roulette (virtual_packet.py:221-226)
                if (P.debug_flags & 67108864) vtraced += 1ull << 40;  // tests: v-packets decided on the prefix sums -> counters[7] >> 40
            }
            else if (screened < 0) err = screened;
            else {
                double tau_v;
                err = vp_trace<FULL, G>(P, rng, dr, p.r, v_mu, v_nu, v_energy, p.shell, p.next_line_id, tau_v, my_visits, geo);
                if (!err) v_energy *= mcm::exp(-tau_v);
            }
            if (dr.overflow) err = ERR_UNSUPPORTED;
        }
        // the first lane whose draw consumption differs from the prediction (or that failed) ends the validity of the round:
        // it started from the right stream position itself, the lanes after it did not
        const unsigned bad = (unsigned)((__ballot(active && (dr.used != pred_self || err != 0)) >> gshift) & GMASK);
        const int f = bad ? __builtin_ctz(bad) : G;
        const int n_ok = min(n_round, f + 1);
        const int used_f = gbcast<G>(dr.used, f & (G - 1));
        const int extra_f = gbcast<G>(extra_before, f & (G - 1));
        const int err_f = gbcast<G>(err, f & (G - 1));
        if (f < G && err_f) { err_out = err_f; break; }
        vtraced += my_visits;
        if (j < n_ok) {
            ++vcount;
            vvisits += my_visits;
            // add_vpacket_collection_to_histogram (modes/montecarlo_transport.py:166-195)
            if (!(v_nu < P.grid0 || v_nu > P.grid_last) && v_energy != 0.0) {  // (a dropped v-packet adds 0.0: the bin keeps its bits)
                long long idx = (long long)floor((v_nu - P.grid0) / P.delta_nu);
                atomic_add_f64(&P.vhist[idx], v_energy);
            }
            if (C->vlog_count) {
                unsigned long long slot = gatomic_add_u64(C->vlog_count, 1ull);
                if ((long long)slot < C->vlog_capacity) {
                    glob(C->vlog_packet)[slot] = packet_index; glob(C->vlog_seq)[slot] = vseq + j;
                    glob(C->vlog_nu)[slot] = v_nu; glob(C->vlog_energy)[slot] = v_energy; glob(C->vlog_mu)[slot] = v_mu0; glob(C->vlog_r)[slot] = p.r;
                }
            }
        }
        // stream words consumed by the committed v-packets: one mu draw each + their roulette draws
        int extra_used;
        if (f < n_round) extra_used = extra_f + used_f;                                            // lanes < f matched the prediction
        else extra_used = __popc((pred_bits >> done) & ((n_round >= 32) ? 0xffffffffu : ((1u << n_round) - 1u)));  // all matched
        // learn from every trace of this round, valid or not (same mu bin => same optical depth to first order)
        const unsigned obs = (unsigned)((__ballot(active && dr.used > 0) >> gshift) & GMASK);
        const unsigned seen = (n_round >= 32) ? 0xffffffffu : ((1u << n_round) - 1u);
        pred_bits = (pred_bits & ~(seen << done)) | (obs << done);
        vseq += n_ok;
        done += n_ok;
        rng.advance(2 * (n_ok + extra_used));
        rng.draws += n_ok + extra_used;
    }
    return err_out;
}

template <int G, int BLOCK>
__host__ __device__ constexpr size_t group_kernel_lds_bytes(int n_shells)
{
    return (size_t)(BLOCK / G) * sizeof(LdsTracker) + 6 * (size_t)n_shells * sizeof(double);  // J, nu_bar; r_inner, r_outer, n_e, tau row sums for the v-packet traces
}

template <bool FULL, bool TRACK, int G, int BLOCK, int OCC, bool VPK>
__global__ void __launch_bounds__(BLOCK, OCC) propagate_group_kernel(GroupArgs P, uint32_t *__restrict__ seeded_states,
                                                                long long chunk_first, long long chunk_count)
{
    constexpr int NGROUPS = BLOCK / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    LdsTracker *lds_trk = reinterpret_cast<LdsTracker *>(lds_raw);    // [NGROUPS]
    double *lds_J = reinterpret_cast<double *>(lds_raw + NGROUPS * sizeof(LdsTracker));
    double *lds_nubar = lds_J + P.n_shells;
    for (int s = threadIdx.x; s < 2 * P.n_shells; s += BLOCK) lds_J[s] = 0.0;
    double *lds_geo = lds_nubar + P.n_shells;  // r_inner | r_outer | n_e: the first link of every shell crossing's chain of look-ups
    if (VPK)
        for (int s = threadIdx.x; s < P.n_shells; s += BLOCK) {
            lds_geo[s] = P.r_inner[s]; lds_geo[P.n_shells + s] = P.r_outer[s]; lds_geo[2 * P.n_shells + s] = P.n_e[s];
            lds_geo[3 * P.n_shells + s] = P.tau_rowsum ? P.tau_rowsum[s] : 0.0;
        }
    __syncthreads();

    const int j = threadIdx.x & (G - 1);
    const int g = threadIdx.x / G;
    const int lane = threadIdx.x & 63;
    const int copy = P.n_est_copies > 1 ? (xcc_id() % P.n_est_copies) : 0;
    double *jb = P.jblue_t + (size_t)copy * P.est_copy_stride;
    double *ed = P.edot_t + (size_t)copy * P.est_copy_stride;
    const double t = P.t_exp;

    GroupRng<G> rng;
    rng.attach(seeded_states);
    rng.draws = 0;
    LdsTracker &trk = lds_trk[g];
    GroupCounters cn;
    unsigned long long draws_total = 0;
    unsigned vvisits = 0, vcount = 0;  // per-lane v-packet work counters
    unsigned long long vtraced = 0;    // line visits of all v-packet traces incl. discarded speculative ones
    int vseq = 0;                      // v-packets emitted so far by this group's packet (group-uniform)
    unsigned pred_bits = 0;            // roulette predictor of the v-packet volleys (see volley_group)
    bool want_volley = false;          // this group's packet is paused until the wave's next volley phase

    // wave-level packet batches: PACKET_BATCH indices are reserved per global atomic and handed to the wave's groups.
    // batch_next / batch_end / exhausted are wave-uniform and only modified in wave-uniform control flow.
    long long batch_next = 0, batch_end = 0;
    bool exhausted = false;
    Packet p;
    p.status = ST_EMITTED;  // "needs a packet"
    p.r = p.mu = p.nu = p.energy = 0.0; p.shell = 0; p.next_line_id = 0;
    long long pkt = -1;
    bool done = false;      // this group has no more work

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
                    // wave-uniform control flow: lane 0 is the first active lane, so this makes the batch scalar
                    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
                    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
                    base = ((unsigned long long)bhi << 32) | blo;
                    batch_next = min((long long)base, chunk_count);
                    batch_end = min((long long)base + PACKET_BATCH, chunk_count);
                    if (batch_next >= batch_end) { exhausted = true; break; }
                }
                const int take = (int)min((long long)(n_want - served), batch_end - batch_next);
                if (need && j == 0 && rank >= served && rank < served + take) mine = batch_next + (rank - served);
                batch_next += take;
                served += take;
            }
            mine = __shfl(mine, 0, G);
            if (need) {
                if (mine < 0) done = true;
                else {
                    pkt = mine;
                    draws_total += (unsigned long long)rng.draws;
                    rng.attach(seeded_states + (size_t)pkt * MT_N);
                    rng.draws = 0;
                    const long long i = chunk_first + pkt;
                    asm volatile("" ::: "memory");  // keep the cold-argument loads inside this (rare) branch
                    const DeviceProblem *C = P.cold;
                    p.r = glob(C->r0)[i]; p.mu = glob(C->mu0)[i]; p.nu = glob(C->nu0)[i]; p.energy = glob(C->e0)[i];
                    p.shell = 0; p.status = ST_IN_PROCESS;
                    if (TRACK && j == 0) {
                        const double nan = __builtin_nan("");
                        trk.radius = trk.nu = trk.energy = nan;
                        trk.before_nu = trk.before_mu = trk.before_energy = nan;
                        trk.after_nu = trk.after_mu = trk.after_energy = nan;
                        trk.shell_id = -1; trk.interaction_type = -1; trk.line_absorb_id = -1; trk.line_emit_id = -1;
                        trk.interactions_count = 0;
                        trk.boundary_buffer = 0;  // -1 + the initial track_boundary_event
                    }
                    {   // set_packet_props_{partial,full}_relativity (classic/packet_propagation.py:254-318)
                        double velocity = p.r / t;
                        double inv = inverse_doppler_factor<FULL>(velocity, p.mu);
                        if (FULL) {
                            double beta = velocity / C_LIGHT;
                            p.nu *= inv; p.energy *= inv;
                            p.mu = (p.mu + beta) / (1 + beta * p.mu);
                        } else { p.nu *= inv; p.energy *= inv; }
                    }
                    {   // initialize_line_id (packets/radiative_packet.py:96-110)
                        double velocity = p.r / t;
                        double comov_nu = p.nu * doppler_factor<FULL>(velocity, p.mu);
                        // number of lines with nu_line >= comov_nu (searchsorted on the reversed list); the bucket index
                        // narrows it to the few lines sharing comov_nu's key, which are compared exactly
                        int lo;
                        {
                            const long long kk = (long long)((unsigned long long)__double_as_longlong(comov_nu > 0.0 ? comov_nu : 0.0) >> P.bucket_shift) - P.bucket_kmin;
                            if (kk >= P.bucket_n) lo = 0;
                            else if (kk < 0) lo = P.n_lines;
                            else {
                                lo = P.bucket_first[kk];
                                const int hi = kk > 0 ? P.bucket_first[kk - 1] : P.n_lines;
                                while (lo < hi && P.nu_line[(unsigned)lo] >= comov_nu) ++lo;
                            }
                        }
                        if (lo == P.n_lines) lo -= 1;
                        p.next_line_id = lo;
                    }
                    if (VPK) {  // volley at launch (classic/packet_propagation.py:109-118), run in the next volley phase
                        vseq = 0;
                        pred_bits = P.tau_pfx ? 0xffffffffu : 0u;  // (screening on: a new packet's first volley is predicted dropped, i.e. screened, too -- see propagate_wave.hpp)
                        want_volley = true;
                    }
                }
            }
        }
        if (__ballot(!done) == 0ull) break;
        if (VPK) {
            // Volley phase.  A volley keeps n_v lanes of ONE group busy for a long time, so it only runs when every group
            // of the wave that still has a packet is waiting for one: groups that reach an interaction early pause (cheap:
            // the others are stepping through boundary crossings) instead of making the others idle through a volley.
            const bool live = !done && p.status == ST_IN_PROCESS;
            const unsigned long long live_mask = __ballot(live), want_mask = __ballot(live && want_volley);
            if (want_mask != 0ull && want_mask == live_mask) {
                if (live && want_volley) {
                    const int verr = volley_group<FULL, G>(P, p, rng, j, chunk_first + pkt, vseq, pred_bits, vvisits, vcount, vtraced, lds_geo);
                    want_volley = false;
                    if (verr) {
                        if (j == 0) {
                            const DeviceProblem *C2 = P.cold;
                            gatomic_min_i64(&C2->first_error[0], chunk_first + pkt);
                            glob(C2->out_nu)[chunk_first + pkt] = (double)verr;
                            glob(C2->out_e)[chunk_first + pkt] = -99.0;
                        }
                        p.status = ST_EMITTED;
                    }
                }
            }
            if (want_volley) continue;  // paused until the wave's volley phase
        }
        if (done || p.status != ST_IN_PROCESS) continue;  // (a volley may have failed: fetch another packet)

        // ---------------------------------------------------------------- one event of this group's packet
        double velocity = p.r / t;
        double dop = doppler_factor<FULL>(velocity, p.mu);
        double chi_e = P.n_e[p.shell] * P.sigma_thomson;
        if (FULL) chi_e *= dop;
        double distance;
        int type = 0, delta = 0;
        int err = trace_packet_group<FULL, G>(P, p, rng, j, chi_e, P.r_inner[p.shell], P.r_outer[p.shell], dop, jb, ed, distance, type, delta, cn);
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
                if (TRACK && j == 0) trk.boundary_buffer += 1;
                cross_shell(p.shell, p.status, delta, P.n_shells);
            } else {
                const double before_nu = p.nu, before_mu = p.mu, before_energy = p.energy;
                const int absorb = (type == IT_LINE) ? p.next_line_id : -1;
                int emit_id = -1;
                // common part of line_scatter_event (interaction_event_callers.py:187-239) and thomson_scatter
                // (interaction_events.py:184-217): Doppler with the old angle, new isotropic angle, Doppler back
                double vel = p.r / t;
                double old_dop = doppler_factor<FULL>(vel, p.mu);
                double comov_nu = p.nu * old_dop;
                double comov_energy = p.energy * old_dop;
                p.mu = 2.0 * rng.random(j) - 1.0;
                double inv_new = inverse_doppler_factor<FULL>(vel, p.mu);
                p.energy = comov_energy * inv_new;
                if (type == IT_LINE) {
                    int emit = p.next_line_id;
                    if (P.line_interaction_type != 0)
                    {
                        const int2 blk = P.line_block[(unsigned)p.next_line_id];
                        err = macro_atom_group<G>(P, rng, j, blk.x, blk.y, p.shell, emit, cn);
                    }
                    if (!err) {  // line_emission (interaction_events.py:227-258); its inverse Doppler factor == inv_new
                        p.nu = P.nu_line[emit] * inv_new;
                        p.next_line_id = emit + 1;
                        emit_id = emit;
                    }
                } else {
                    p.nu = comov_nu * inv_new;
                }
                if (FULL && !err) p.mu = aberration_cmf_to_lf(p.r, t, p.mu);
                if (TRACK && j == 0 && !err) {
                    trk.before_nu = before_nu; trk.before_mu = before_mu; trk.before_energy = before_energy;
                    trk.line_absorb_id = absorb; trk.line_emit_id = emit_id;
                    trk.after_nu = p.nu; trk.after_mu = p.mu; trk.after_energy = p.energy;
                    trk.interactions_count += 1 + trk.boundary_buffer;
                    trk.boundary_buffer = 0;
                    trk.radius = p.r; trk.nu = p.nu; trk.energy = p.energy; trk.shell_id = p.shell;
                    trk.interaction_type = type;
                }
                // volley after a line or electron-scattering interaction (classic/packet_propagation.py:201-244)
                if (VPK && !err) want_volley = true;
            }
        }
        if (err) {
            const long long i = chunk_first + pkt;
            if (j == 0) {
                asm volatile("" ::: "memory");
                const DeviceProblem *C = P.cold;
                gatomic_min_i64(&C->first_error[0], i);
                glob(C->out_nu)[i] = (double)err;
                glob(C->out_e)[i] = -99.0;
            }
            p.status = ST_EMITTED;
        } else if (p.status != ST_IN_PROCESS) {
            // set_packet_collection_output (modes/montecarlo_transport.py:70-90)
            const long long i = chunk_first + pkt;
            if (j == 0) {
                asm volatile("" ::: "memory");  // keep the cold-argument loads inside this (rare) branch
                const DeviceProblem *C = P.cold;
                glob(C->out_nu)[i] = p.nu;
                glob(C->out_e)[i] = (p.status == ST_REABSORBED) ? -p.energy : p.energy;
                if (TRACK) {
                    glob(C->li_radius)[i] = trk.radius; glob(C->li_nu)[i] = trk.nu; glob(C->li_energy)[i] = trk.energy;
                    glob(C->li_before_nu)[i] = trk.before_nu; glob(C->li_before_mu)[i] = trk.before_mu; glob(C->li_before_energy)[i] = trk.before_energy;
                    glob(C->li_after_nu)[i] = trk.after_nu; glob(C->li_after_mu)[i] = trk.after_mu; glob(C->li_after_energy)[i] = trk.after_energy;
                    glob(C->li_shell_id)[i] = trk.shell_id; glob(C->li_interaction_type)[i] = trk.interaction_type;
                    glob(C->li_line_absorb_id)[i] = trk.line_absorb_id; glob(C->li_line_emit_id)[i] = trk.line_emit_id;
                    glob(C->li_interactions_count)[i] = trk.interactions_count;
                }
            }
        }
    }
    draws_total += (unsigned long long)rng.draws;
    __syncthreads();
    const DeviceProblem *C = P.cold;
    for (int s = threadIdx.x; s < P.n_shells; s += BLOCK) {
        if (lds_J[s] != 0.0) gatomic_add_f64(&C->J[s], lds_J[s]);
        if (lds_nubar[s] != 0.0) gatomic_add_f64(&C->nubar[s], lds_nubar[s]);
    }
    if (j == 0) {
        gatomic_add_u64(&C->counters[0], cn.visits);
        gatomic_add_u64(&C->counters[1], (unsigned long long)cn.events);
        gatomic_add_u64(&C->counters[2], (unsigned long long)cn.macro);
        gatomic_add_u64(&C->counters[5], draws_total);
    }
    if (VPK) {
        if (vvisits) gatomic_add_u64(&C->counters[3], (unsigned long long)vvisits);
        if (vcount) gatomic_add_u64(&C->counters[4], (unsigned long long)vcount);
        if (vtraced) gatomic_add_u64(&C->counters[7], vtraced);
    }
}

}  // namespace mc
