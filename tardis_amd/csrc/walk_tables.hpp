// walk_tables.hpp -- compact tables of the macro-atom walk (macroatom mode of the wave kernel).
//
// macro_atom_interaction (tardis/transport/montecarlo/macro_atom.py:52-104) adds the probabilities of the activated level's
// block up until the sum exceeds the drawn number xi; cum_t holds those running sums (same additions, same order), so a jump
// selects the first entry of the block with cum > xi.  Measured on MI355X (profiles/r02_memory_ceilings.txt): the walk is bound
// by the number of memory requests that leave the L2 -- the chip retires ~55 G random requests/s of up to 64 bytes when every
// LANE has its own request in flight, but only ~20 G/s of the 128-byte lines a 16-lane group reads together -- and at
// 20 shells x 1.5e6 transitions x 8 bytes nothing of cum_t stays in a cache.  So every lane walks for its own packet, on
// tables sized for one small request per step:
//
//   cum16[s][c]   = min(65535, floor(cum_t * 65536)) as u16: the block of level b starts at the 16-byte aligned compact index
//                   c0(b) and is padded to a multiple of 8 entries with 0xffff.  With x = floor(xi * 65536):
//                       cum16 > x  =>  cum > xi          cum16 < x  =>  cum <= xi          (floor is monotone)
//                   so only an entry with cum16 == x (2^-16 of the draws per entry) needs the fp64 sum: same decisions as the
//                   reference, 2 bytes per transition instead of 8.  A block of <= 32 transitions is one 16..64-byte read.
//   rec16[c]      = what the selected transition leads to, shell-independent, one 16-byte read: {c0, rows} of the destination
//                   level's block, or {line id, EMIT, nu of that line} for an emission (transition type -1), or
//                   {0, UNSUPPORTED} for the reference's other negative types (continuum processes, not in the classic mode).
//   quad_info[q]  = {original transition index of compact entry 8 q, entries from there to the end of its block}: maps a
//                   compact index back to cum_t for the exact comparison.
//   line_block_c[line] = {c0, rows} of the block the line activates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

constexpr unsigned WALK_EMIT = 0x80000000u, WALK_UNSUPPORTED = 0x40000000u;
struct __attribute__((aligned(16))) WalkRec {
    unsigned a, b;  // internal transition: compact start and rows of the destination block; emission: line id, WALK_EMIT [| WALK_UNSUPPORTED]
    double nu;      // emission: frequency of the line (line_emission, interaction_events.py:227-258, needs it next)
};
constexpr int WALK_WINDOW_QUADS = 4;   // 8-entry quads (16 bytes) a lane reads per jump
constexpr int WALK_SLACK = 8 * WALK_WINDOW_QUADS;  // entries of slack at the end of every cum16 row

// one thread per (shell, quad): the 8 u16 entries of a quad from the fp64 running sums
__global__ void __launch_bounds__(256) walk_cum16_kernel(const double *__restrict__ cum_t, const int2 *__restrict__ quad_info,
                                                          long long n_quads, long long n_trans, int n_shells, unsigned stride,
                                                          unsigned short *__restrict__ cum16)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads * n_shells) return;
    const long long q = i % n_quads;
    const int s = (int)(i / n_quads);
    const int2 info = quad_info[q];
    const double *c = cum_t + (long long)s * n_trans + info.x;
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned e = 0xffffu;  // padding: never below, never equal (except x = 65535, which takes the exact path)
        if (k < info.y) {
            const double t = c[k] * 65536.0;  // exact scaling
            e = t >= 65535.0 ? 65535u : (unsigned)t;  // (probabilities are >= 0 here: negative ones keep the problem off this walk)
        }
        v[k] = e;
    }
    uint4 out;
    out.x = v[0] | (v[1] << 16); out.y = v[2] | (v[3] << 16); out.z = v[4] | (v[5] << 16); out.w = v[6] | (v[7] << 16);
    reinterpret_cast<uint4 *>(cum16 + (size_t)s * stride)[q] = out;
}

// packed u16 counting: for the two entries of a dword, +1 in the respective half of `less` where entry < x and of `gt`
// where entry > x (xx = x | x << 16)
typedef unsigned short walk_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void walk_count(unsigned d, unsigned xx, unsigned &less, unsigned &gt)
{
    const walk_u16x2 dv = __builtin_bit_cast(walk_u16x2, d), xv = __builtin_bit_cast(walk_u16x2, xx);
    const walk_u16x2 one = {1, 1};
    walk_u16x2 l = __builtin_elementwise_sub_sat(xv, dv);  // > 0 where entry < x
    walk_u16x2 g = __builtin_elementwise_sub_sat(dv, xv);  // > 0 where entry > x
    l = __builtin_elementwise_min(l, one);
    g = __builtin_elementwise_min(g, one);
    less = __builtin_bit_cast(unsigned, (walk_u16x2)(__builtin_bit_cast(walk_u16x2, less) + l));
    gt = __builtin_bit_cast(unsigned, (walk_u16x2)(__builtin_bit_cast(walk_u16x2, gt) + g));
}
__device__ __forceinline__ void walk_count_quad(const uint4 &w, unsigned xx, unsigned &less, unsigned &gt)
{
    walk_count(w.x, xx, less, gt); walk_count(w.y, xx, less, gt); walk_count(w.z, xx, less, gt); walk_count(w.w, xx, less, gt);
}

}  // namespace mc
