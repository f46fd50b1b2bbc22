// packet_source.hpp -- black-body packet source on the device.
//
// Reproduces BlackBodySimpleSource.create_packets (tardis/transport/montecarlo/packet_source/base.py:195-253,
// black_body.py:140-222) including NumPy's Generator(PCG64) streams, so that the packets a host run would have
// sampled with np.random.default_rng(base_seed + seed_offset) are produced in HBM, in place, with no PCIe traffic:
//
//   packet_seeds = rng.choice(MAX_SEED_VAL, n, replace=True)   -> bounded Lemire 32-bit draws (with rejection)
//   xis          = rng.random((5, n))                          -> u64 draws  q0 + k n + i   (row-major)
//   mus          = sqrt(rng.random(n))                         -> u64 draws  q0 + 5 n + i
//
// PCG64 (XSL-RR 128/64) is a 128-bit LCG, so any draw is reachable by jump-ahead: every thread jumps straight to its
// packet's positions (jump tables of the 2^j-step affine maps are precomputed on the host), which makes the source
// embarrassingly parallel and lets a rank generate only its shard [first, first + count) of a global n-packet draw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_math.hpp"

namespace mc {

typedef unsigned __int128 u128;

struct PcgAffine {  // x -> mult * x + plus  (mod 2^128)
    uint64_t mult_lo, mult_hi, plus_lo, plus_hi;
};

constexpr int PCG_JUMP_BITS = 48;

struct PacketSourceArgs {
    long long n_total, first, count;  // global draw size, this shard
    uint64_t state_lo, state_hi;      // PCG64 state at the start of the draw (after seeding)
    const PcgAffine *jump;            // [PCG_JUMP_BITS] 2^j-step maps
    PcgAffine step_n;                 // n_total-step map (row stride of xis)
    uint32_t seed_range_excl;         // MAX_SEED_VAL: seeds are drawn from [0, seed_range_excl)
    uint32_t seed_threshold;          // Lemire rejection threshold, (2^32 - range) % range
    const long long *rejected;        // sorted u32 positions of rejected draws
    int n_rejected;
    long long xi_first_u64;           // u64 draw index where rng.random((5, n)) starts
    const double *l_array;            // cumulative sum of k^-4, k = 1 .. l_samples-1
    int n_l;
    double l_coef, radius, kT, h, energy;
    double *r0, *mu0, *nu0, *e0;
    uint32_t *seeds;
};

__device__ __forceinline__ u128 make128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

__device__ __forceinline__ u128 pcg_apply(const PcgAffine &a, u128 s)
{
    return make128(a.mult_hi, a.mult_lo) * s + make128(a.plus_hi, a.plus_lo);
}

// state after `delta` steps
__device__ __forceinline__ u128 pcg_jump(u128 s, unsigned long long delta, const PcgAffine *jump)
{
    for (int j = 0; delta; ++j, delta >>= 1)
        if (delta & 1) s = pcg_apply(jump[j], s);
    return s;
}

__device__ __forceinline__ uint64_t pcg_output(u128 s)
{  // XSL-RR 128/64
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

__device__ __forceinline__ double pcg_double(uint64_t u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }

// u32 draw number `pos` of the stream that starts at `s0`: numpy hands out the low half of a fresh u64 first, then
// the high half (pcg64_next32)
__device__ __forceinline__ uint32_t pcg_u32_at(u128 s0, long long pos, const PcgAffine *jump)
{
    const uint64_t u = pcg_output(pcg_jump(s0, (unsigned long long)(pos >> 1) + 1, jump));
    return (pos & 1) ? (uint32_t)(u >> 32) : (uint32_t)u;
}

// pass 1: positions of the u32 draws the bounded-integer sampler rejects (probability threshold / 2^32 each)
__global__ void packet_source_scan_kernel(PacketSourceArgs a, long long n_positions, long long *rejected, int capacity,
                                          unsigned int *n_rejected)
{
    const u128 s0 = make128(a.state_hi, a.state_lo);
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; 2 * q < n_positions;
         q += (long long)gridDim.x * blockDim.x) {
        const uint64_t u = pcg_output(pcg_jump(s0, (unsigned long long)q + 1, a.jump));
        const uint32_t w[2] = {(uint32_t)u, (uint32_t)(u >> 32)};
        for (int h = 0; h < 2; ++h) {
            const long long pos = 2 * q + h;
            if (pos >= n_positions) break;
            const uint32_t leftover = (uint32_t)((uint64_t)w[h] * a.seed_range_excl);
            if (leftover < a.seed_threshold) {
                const unsigned int slot = atomicAdd(n_rejected, 1u);
                if ((int)slot < capacity) rejected[slot] = pos;
            }
        }
    }
}

// pass 2: one thread per packet
__global__ void packet_source_kernel(PacketSourceArgs a)
{
    const u128 s0 = make128(a.state_hi, a.state_lo);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < a.count; j += (long long)gridDim.x * blockDim.x) {
        const long long i = a.first + j;  // global packet index
        // seed: the i-th accepted u32 draw
        long long pos = i;
        for (int k = 0; k < a.n_rejected && a.rejected[k] <= pos; ++k) ++pos;
        const uint32_t w = pcg_u32_at(s0, pos, a.jump);
        a.seeds[j] = (uint32_t)(((uint64_t)w * a.seed_range_excl) >> 32);
        // xis[k][i], k = 0..4, then the mu draw: six u64 outputs n_total apart
        u128 s = pcg_jump(s0, (unsigned long long)(a.xi_first_u64 + i) + 1, a.jump);
        const double xi0 = pcg_double(pcg_output(s));
        double prod = 1.0;
        for (int k = 1; k < 5; ++k) {
            s = pcg_apply(a.step_n, s);
            const double xi = pcg_double(pcg_output(s));
            prod = (k == 1) ? xi : prod * xi;  // np.prod(xis[1:], 0): ((x1 x2) x3) x4
        }
        s = pcg_apply(a.step_n, s);
        const double z = pcg_double(pcg_output(s));
        // l_min = l_array.searchsorted(xi0 * l_coef) + 1 (side='left': number of entries < v)
        const double v = xi0 * a.l_coef;
        int lo = 0, hi = a.n_l;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.l_array[mid] < v) lo = mid + 1; else hi = mid;
        }
        const double l_min = (double)lo + 1.0;
        const double x = -mcm::log(prod) / l_min;
        a.nu0[j] = x * a.kT / a.h;
        a.mu0[j] = sqrt(z);
        a.r0[j] = a.radius;
        a.e0[j] = a.energy;
    }
}

}  // namespace mc
