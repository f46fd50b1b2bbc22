// mc_device.hpp -- device-side data model and leaf physics of the MC packet engine (gfx950).
//
// Scalar per-packet physics written to reproduce the reference's floating-point operation order exactly
// (IEEE double, no FMA contraction: the translation unit is compiled with -ffp-contract=off), so that the
// per-packet results are bit-identical to the CPU oracle.  Reference lines are cited at each function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_math.hpp"

namespace mc {

constexpr double C_LIGHT = 2.99792458e10;       // tardis/constants.py:1 (CODATA 2010)
constexpr double CLOSE_LINE_THRESHOLD = 1e-14;  // configuration/constants.py:4
constexpr double MISS_DISTANCE = 1e99;          // configuration/constants.py:6

enum : int { IT_BOUNDARY = 1, IT_LINE = 2, IT_ESCATTERING = 4 };
enum : int { ST_IN_PROCESS = 0, ST_EMITTED = 1, ST_REABSORBED = 2 };
enum : int { ERR_MONTECARLO = -3, ERR_MACRO_ATOM = -4, ERR_UNSUPPORTED = -5 };

constexpr int MT_N = 624;

// ---- deferred line-estimator accumulation (estimator_log.hpp): one record per trace
// In partial relativity the term update_line_estimators adds for a visited line, energy * (1 - (d_line + mu r) / (t c)),
// is energy * nu_line / nu exactly (d_line is where the packet's comoving frequency equals nu_line), i.e. a per-trace
// constant times nu_line; in full relativity it is the energy itself.  The record therefore carries the two constants
// (Edotlu and j_blue) and the range of lines; the factor nu_line is applied once per line when a tile is flushed.  The sums
// agree with the reference's to rounding (<= ~1e-14 relative per term: the reference's own rounding and its d = 0 for
// lines closer than 1e-14), far inside the summation-order tolerance the estimators are compared with.
struct __attribute__((aligned(8))) LineVisitRecord {
    double c_e, c_jb;   // Edotlu / j_blue constants of the trace (energy / nu, energy / nu^2; full relativity: energy, energy / nu)
    unsigned idx0;      // shell * n_lines + first line visited
    unsigned n;         // number of lines visited
};
static_assert(sizeof(LineVisitRecord) == 24, "record layout");

struct EstimatorLog {
    // The log is a pool of equal chunks (`region_capacity` records each); a wave appends to the chunk it holds (no atomics, no empty
    // slots) and takes the next free one from the pool -- one atomic per chunk -- when that is full.  An epoch ends when the pool is
    // empty: every wave then suspends within one chunk's worth of passes of the others.  (Rounds 1-3 gave every wave ONE region
    // of capacity / waves records: the waves filled theirs at different times, and from the first suspension to the last -- 10-20 %
    // of an epoch -- a growing part of the chip sat idle: 0.27 s of a 3.6-s step, profiles/r04_estimator_cost.txt.)
    LineVisitRecord *records;          // [n_regions][region_capacity]
    unsigned *keys;                    // bin of each record, same layout
    unsigned *region_count;            // [n_regions] records written to each chunk (stored when its wave leaves it)
    unsigned *pool_next;               // next free chunk of the pool
    unsigned region_capacity;          // records per chunk; 0: no log, the kernels add their terms directly
    int n_regions;                     // chunks of the pool
    int tiles_per_shell;
};

// Everything a propagation kernel needs, passed by value (kernarg segment -> SGPRs).
struct DeviceProblem {
    // packets (SoA, coalesced by packet index)
    long long n_packets;
    const double *r0, *mu0, *nu0, *e0;
    const uint32_t *seeds;
    double *out_nu, *out_e;
    // last-interaction tracker SoA (all null when tracking is off)
    double *li_radius, *li_nu, *li_energy, *li_before_nu, *li_before_mu, *li_before_energy, *li_after_nu,
        *li_after_mu, *li_after_energy;
    long long *li_shell_id, *li_interaction_type, *li_line_absorb_id, *li_line_emit_id, *li_interactions_count;
    // wave kernel: one 64-byte record per packet, rewritten at every interaction (TrackerRecord, propagate_wave.hpp) and
    // unpacked into the arrays above once the propagation is over -- one write request per interaction instead of nine
    uint4 *li_rec;
    // geometry
    int n_shells;
    const double *r_inner, *r_outer;
    double t_exp;
    // opacity (shell-major tables)
    int n_lines, n_trans;
    const double *nu_line;  // [L] descending
    const double *tau_t;    // [S][L]
    const double *n_e;      // [S]
    const double *prob_t;   // [S][T]
    const int *line2level;  // [L]
    const int *block_edge;  // [levels+1]
    const int *ttype;       // [T]
    const int *dest;        // [T]
    const int *tline;       // [T]
    // estimators (shell-major), n_est_copies private copies selected by XCC id
    double *J, *nubar;      // [S]
    double *jblue_t, *edot_t;  // [copies][S][L]
    long long est_copy_stride; // S*L
    int n_est_copies;
    double *vhist;          // [G]
    // config
    int line_interaction_type, disable_line_scattering;
    long long n_vpackets;
    double survival_probability, tau_russian, spawn_start, spawn_end, sigma_thomson;
    const double *grid;
    int n_grid;
    double grid0, grid_last, delta_nu;
    // v-packet log (optional)
    unsigned long long *vlog_count;
    long long vlog_capacity;
    long long *vlog_packet;
    int *vlog_seq;
    double *vlog_nu, *vlog_energy, *vlog_mu, *vlog_r;
    // scratch / bookkeeping
    uint32_t *rng_state;            // [n_slots][624]
    unsigned long long *counters;   // [8]
    long long *first_error;         // {packet index (min), code}
    unsigned long long *next_packet;  // work counter for persistent scheduling
    int debug_flags;                  // profiling experiments only: 1 = skip j_blue/Edotlu atomics, 2 = skip J/nu_bar
    EstimatorLog log;                 // line-visit log of the cooperative kernels (capacity 0: they add their terms directly)
};

// Slim by-value argument block of the cooperative kernel: only what the inner loops touch stays in SGPRs; everything
// used once per packet (inputs, outputs, tracker arrays, counters) is read through `cold` (a device copy of the full
// DeviceProblem) at the point of use.
struct WalkRec;
struct GroupArgs {
    const DeviceProblem *cold;
    int n_shells, n_lines, n_trans;
    int line_interaction_type, disable_line_scattering, debug_flags, n_est_copies;
    double t_exp, sigma_thomson;
    double tc, rcp_tc;  // t_exp * c and its correctly rounded reciprocal (host-computed, kernel-uniform)
    const double *r_inner, *r_outer, *nu_line, *tau_t, *n_e, *prob_t;
    // macro atom, packed for one dependent load per jump: line_block[line] = {first, end} transition of the level the
    // line activates; trans_rec[t] = {emission line id, transition type, first, end transition of the destination level}
    const int2 *line_block;
    const int4 *trans_rec;
    // wave kernel: cum_t[s][t] = the reference's running sum of the transition probabilities from the start of t's block
    // up to and including t (same additions in the same order, so the jump search compares the very numbers the
    // reference's loop does); trans_nu[t] = nu of the line transition t emits (0 for internal transitions)
    const double *cum_t, *trans_nu;
    // compact tables of the per-lane macro-atom walk (walk_tables.hpp; null unless the launch uses them)
    const unsigned short *cum16;  // [S][cum16_stride]
    const struct WalkRec *rec16;  // [compact transitions]
    const int2 *quad_info;        // [compact transitions / 8]
    unsigned cum16_stride;
    const unsigned *hot_sec;      // [S][blocks][16]: hot sectors (null: no block is entered through one)
    const int2 *blk_tab;          // [blocks] {compact start, rows}
    unsigned hot_stride;          // dwords per shell of hot_sec (16 * blocks)
    double *jblue_t, *edot_t;
    long long est_copy_stride;
    unsigned long long *next_packet;
    // v-packets (only read by the VPK instantiations)
    long long n_vpackets;
    double survival_probability, tau_russian, spawn_start, spawn_end, grid0, grid_last, delta_nu;
    double *vhist;
    // frequency-bucket index of the line list: key(nu) = (bits(nu) >> bucket_shift) - bucket_kmin is monotone in nu;
    // bucket_first[k] = index of the first line (descending list) whose key is <= k
    const int *bucket_first;
    int bucket_shift, bucket_n;
    long long bucket_kmin;
    // v-packet screening (tau_prefix.hpp): prefix sums of tau along every shell's row, [S][L + 1], and the row totals; null when
    // the screening is off (survival probability > 0, a negative optical depth, debug flag)
    const double *tau_pfx, *tau_rowsum;
    // interleaved sweep table of the lane sweeps (propagate_wave_kernel<..., NT>): nt_t[shell * nt_stride + line] = {nu_line[line], tau_t[shell][line]};
    // null / 0 unless the launch uses it
    const double *nt_t;
    unsigned nt_stride;
};

struct Packet {
    double r, mu, nu, energy;
    int next_line_id, shell, status;
};

// ---- MT19937, numpy legacy stream (np.random.seed / np.random.random; montecarlo_transport.py:65).
// State lives in global scratch, one contiguous 2496-byte block per lane; values are produced incrementally
// (one tempering + one in-place state update per 32-bit output), which is the same sequence as numpy's
// block-wise regeneration.
struct Rng {
    uint32_t *mt;
    int idx;
    long long draws;

    __device__ __forceinline__ void seed(uint32_t *state, uint32_t s)
    {
        mt = state;
        uint32_t x = s;
        mt[0] = x;
        for (int i = 1; i < MT_N; ++i) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            mt[i] = x;
        }
        idx = 0;
        draws = 0;
    }
    __device__ __forceinline__ uint32_t next_u32()
    {
        int k = idx;
        int k1 = (k + 1 == MT_N) ? 0 : k + 1;
        int km = (k + 397 >= MT_N) ? k + 397 - MT_N : k + 397;
        uint32_t y = (mt[k] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
        uint32_t v = mt[km] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        mt[k] = v;
        idx = k1;
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
    __device__ __forceinline__ double random()
    {
        uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        ++draws;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

// ---- frame transformations (tardis/transport/frame_transformations.py:12-109)
template <bool FULL>
__device__ __forceinline__ double doppler_factor(double velocity, double mu)
{
    const double inv_c = 1 / C_LIGHT;
    double beta = velocity * inv_c;
    if (!FULL) return 1.0 - mu * beta;
    return (1.0 - mu * beta) / sqrt(1 - beta * beta);
}
template <bool FULL>
__device__ __forceinline__ double inverse_doppler_factor(double velocity, double mu)
{
    const double inv_c = 1 / C_LIGHT;
    double beta = velocity * inv_c;
    if (!FULL) return 1.0 / (1.0 - mu * beta);
    return (1.0 + mu * beta) / sqrt(1 - beta * beta);
}
__device__ __forceinline__ double aberration_cmf_to_lf(double r, double t, double mu)
{
    double ct = C_LIGHT * t;
    double beta = r / ct;
    return (mu + beta) / (1.0 + beta * mu);
}
__device__ __forceinline__ double aberration_lf_to_cmf(double r, double t, double mu)
{
    double ct = C_LIGHT * t;
    double beta = r / ct;
    return (mu - beta) / (1.0 - beta * mu);
}

// ---- distances (tardis/transport/geometry/calculate_distances.py:25-112,198-219)
__device__ __forceinline__ void distance_boundary(double r, double mu, double r_inner, double r_outer, double &d,
                                                  int &delta)
{
    if (mu > 0.0) {
        d = sqrt(r_outer * r_outer + ((mu * mu - 1.0) * r * r)) - (r * mu);
        delta = 1;
    } else {
        double check = r_inner * r_inner + (r * r * (mu * mu - 1.0));
        if (check >= 0.0) {
            d = -r * mu - sqrt(check);
            delta = -1;
        } else {
            d = sqrt(r_outer * r_outer + ((mu * mu - 1.0) * r * r)) - (r * mu);
            delta = 1;
        }
    }
}
__device__ __forceinline__ double distance_line_full_relativity(double nu_line, double nu, double t, double r, double mu)
{
    double nu_r = nu_line / nu;
    double ct = C_LIGHT * t;
    return -mu * r + (ct - nu_r * nu_r * sqrt(ct * ct - (1 + r * r * (1 - mu * mu) * (1 + 1.0 / (nu_r * nu_r))))) /
                         (1 + nu_r * nu_r);
}
// returns false on the reference's MonteCarloException("nu difference is less than 0.0")
template <bool FULL>
__device__ __forceinline__ bool distance_line(double nu, double r, double mu, double comov_nu, bool is_last, double nu_line,
                                              double t, double &d)
{
    if (is_last) { d = MISS_DISTANCE; return true; }
    double nu_diff = comov_nu - nu_line;
    double q = nu_diff / nu;
    if (fabs(q) < CLOSE_LINE_THRESHOLD) { d = 0.0; return true; }
    if (!(nu_diff >= 0)) return false;
    if (FULL) d = distance_line_full_relativity(nu_line, nu, t, r, mu);
    else d = q * C_LIGHT * t;
    return true;
}

// a / b with a precomputed y = RN(1/b): q0 = RN(a y), r = a - b q0 (exact, fma), q = RN(q0 + r y) is the correctly
// rounded quotient (Markstein, IBM J. Res. Dev. 34, 1990) as long as a, b, y, q0 are finite and nothing over- or
// underflows; the callers establish that range once per event (div_operands_safe) and otherwise divide normally.
// 3 VALU instead of ~11, bit-identical to a / b (0 mismatches in 4e8 samples incl. adversarial significands on the
// host, 2e6 on the device: tests/test_hip_parity.py::test_exact_division).
template <bool FAST>
__device__ __forceinline__ double exact_div(double a, double b, double y)
{
    if (!FAST) return a / b;
    const double q0 = a * y;
    const double r = __builtin_fma(-q0, b, a);
    return __builtin_fma(r, y, q0);
}
// |x| in [1e-140, 1e140]: products / quotients of two such numbers stay far away from the exponent limits
__device__ __forceinline__ bool mid_range(double x) { double ax = fabs(x); return ax > 1e-140 && ax < 1e140; }

__device__ __forceinline__ void cross_shell(int &shell, int &status, int delta, int n_shells)
{ // packets/movement.py:80-102
    int next = shell + delta;
    if (next >= n_shells) status = ST_EMITTED;
    else if (next < 0) status = ST_REABSORBED;
    else shell = next;
}

// fp64 hardware atomics (global_atomic_add_f64; compile with -munsafe-fp-atomics)
__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- address spaces.  A pointer the kernel LOADED from memory (a field of WaveCold / GroupArgs / DeviceProblem / EstimatorLog
// read through the argument block) is a generic pointer to the compiler, and every access through it becomes a FLAT
// instruction: it counts on the vector-memory AND the LDS counter and returns out of order against LDS accesses, so the
// compiler can neither leave such a load in flight across an LDS access nor wait for "all but N" of them -- every wait turns
// into a full drain (round 3's wave kernel: 189 flat_load against 20 global_load, 75 x `s_waitcnt vmcnt(0) lgkmcnt(0)`).
// All of these pointers are device (global) memory; glob() says so at the point of use (the cast is free, the pointer itself
// still comes out of a scalar load), gload / gstore move whole structures in the widest pieces their alignment allows (class
// types cannot be copied through an address-space-qualified lvalue).
// One trap, measured (profiles/r04_address_space_ab.txt): a load whose address is wave-uniform and provably global becomes a
// SCALAR load; where that put a wave's bookkeeping into SGPRs for the whole kernel the event loop went from 45 to 215 spilled
// VGPRs and lost 25 % -- such loads keep an opaque (vector) index, see the resume code of propagate_wave_kernel.
#define MC_AS1 __attribute__((address_space(1)))
#define MC_G MC_AS1
template <class T> __device__ __forceinline__ MC_G T *glob(T *p) { return (MC_G T *)p; }
template <class T> __device__ __forceinline__ T gload(const T *p)
{
    typedef unsigned gv4 __attribute__((ext_vector_type(4)));
    typedef unsigned gv2 __attribute__((ext_vector_type(2)));
    struct { unsigned w[sizeof(T) / 4]; } u;
    static_assert(sizeof(T) % 4 == 0, "gload: whole dwords");
    constexpr int N = (int)(sizeof(T) / 4);
    if constexpr (alignof(T) >= 16 && N % 4 == 0) {
        const MC_G gv4 *q = (const MC_G gv4 *)p;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { const gv4 t = q[i]; u.w[4 * i] = t.x; u.w[4 * i + 1] = t.y; u.w[4 * i + 2] = t.z; u.w[4 * i + 3] = t.w; }
    } else if constexpr (alignof(T) >= 8 && N % 2 == 0) {
        const MC_G gv2 *q = (const MC_G gv2 *)p;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { const gv2 t = q[i]; u.w[2 * i] = t.x; u.w[2 * i + 1] = t.y; }
    } else {
        const MC_G unsigned *q = (const MC_G unsigned *)p;
#pragma unroll
        for (int i = 0; i < N; ++i) u.w[i] = q[i];
    }
    T r;
    __builtin_memcpy(&r, u.w, sizeof(T));
    return r;
}
template <class T> __device__ __forceinline__ void gstore(T *p, const T &v)
{
    typedef unsigned gv4 __attribute__((ext_vector_type(4)));
    typedef unsigned gv2 __attribute__((ext_vector_type(2)));
    struct { unsigned w[sizeof(T) / 4]; } u;
    static_assert(sizeof(T) % 4 == 0, "gstore: whole dwords");
    __builtin_memcpy(u.w, &v, sizeof(T));
    constexpr int N = (int)(sizeof(T) / 4);
    if constexpr (alignof(T) >= 16 && N % 4 == 0) {
        MC_G gv4 *q = (MC_G gv4 *)p;
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { gv4 t = {u.w[4 * i], u.w[4 * i + 1], u.w[4 * i + 2], u.w[4 * i + 3]}; q[i] = t; }
    } else if constexpr (alignof(T) >= 8 && N % 2 == 0) {
        MC_G gv2 *q = (MC_G gv2 *)p;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { gv2 t = {u.w[2 * i], u.w[2 * i + 1]}; q[i] = t; }
    } else {
        MC_G unsigned *q = (MC_G unsigned *)p;
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = u.w[i];
    }
}
__device__ __forceinline__ void gatomic_add_f64(double *p, double v)
{
    __hip_atomic_fetch_add((MC_AS1 __typeof__(*p) *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long gatomic_add_u64(unsigned long long *p, unsigned long long v)
{
    return __hip_atomic_fetch_add((MC_AS1 __typeof__(*p) *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned gatomic_add_u32(unsigned *p, unsigned v)
{
    return __hip_atomic_fetch_add((MC_AS1 __typeof__(*p) *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gatomic_min_i64(long long *p, long long v)
{
    __hip_atomic_fetch_min((MC_AS1 __typeof__(*p) *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mc
