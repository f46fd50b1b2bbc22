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
//
// Hot sectors (round 4).  A jump costs two dependent requests (running sums, then the record of the selected transition), a
// jump out of a long block a binary search on top -- and real macro-atom blocks are long and skewed: a few transitions carry
// nearly all of a block's probability (A-values times escape probabilities span decades).  For such blocks
//   hot_sec[s][b] = ONE 64-byte sector per (shell, block): the up to six widest intervals [cum(k-1), cum(k)) of the block as
//                   {lo, hi} in 16-bit units, what transition k leads to, and k itself (the reference counts the
//                   transitions it examined):   dwords 0-2 lo x 6 | 3-5 hi x 6 | 6-11 dest x 6 | 12-14 k x 6 | 15 total width
//                   With x = floor(xi * 65536):   lo <= x < hi   =>   cum(k-1) <= xi < cum(k)   (lo = floor(cum(k-1) * 65536) + 1, or 0
//                   for the first transition; hi = floor(cum(k) * 65536), both saturating at 65535: floor(y) + 1 > y and
//                   floor(y) <= y), i.e. transition k is the reference's choice; the order of the entries does not matter.
//                   A number that falls into none of the six (a narrow interval, a tie at an interval's end) is looked up in
//                   the block's own tables as before -- the SAME number, in the next round of the walk.
//                   dest = line id | WALK_EMIT, or block id [| WALK_HOT_DEST when that block is entered through its hot sector].
//   blk_tab[b]    = {c0, rows}: the block's own tables, for the numbers its hot sector does not decide (and for the cold
//                   destinations of hot entries); 8 bytes per block, cache resident.
// A block gets a hot sector when the six intervals cover enough of it on average over the shells (walk_hot_min_mass; blocks of
// more than one window need less: their cold jump starts with a binary search).  line_block_c / rec16 then name the block by
// {block id, -1} / {block id, WALK_HOT} instead of {c0, rows}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

constexpr unsigned WALK_EMIT = 0x80000000u, WALK_UNSUPPORTED = 0x40000000u;
constexpr unsigned WALK_HOT = 0x20000000u;        // rec16[].b of an internal transition: the destination block is entered through its hot sector (a = block id)
constexpr unsigned WALK_HOT_DEST = 0x40000000u;   // dest word of a hot-sector entry: likewise
constexpr int WALK_REDO = 0x40000000;             // parked walk state (rows | WALK_REDO): the number drawn is looked up again, in the block's own tables
constexpr int HOT_ENTRIES = 6;
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

// one thread per (block, shell): the six widest 16-bit intervals of the block's running sums -> its hot sector; `mass` = their
// total width (of 65536).  Only emissions (type -1) and internal transitions (type >= 0) are eligible; `hot_flag` (null in the
// first of the two passes, which only measures) marks the destination blocks that are themselves entered through hot sectors.
__global__ void __launch_bounds__(256) walk_hot_kernel(const double *__restrict__ cum_t, const int *__restrict__ block_edge,
                                                        const int *__restrict__ ttype, const int *__restrict__ dest, const int *__restrict__ tline,
                                                        const unsigned char *__restrict__ hot_flag, int n_blocks, long long n_trans, int n_shells,
                                                        unsigned *__restrict__ hot_sec, unsigned *__restrict__ mass)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_blocks * n_shells) return;
    const int b = (int)(i % n_blocks), s = (int)(i / n_blocks);
    const double *c = cum_t + (long long)s * n_trans;
    const int b0 = block_edge[b], b1 = block_edge[b + 1];
    unsigned lo[HOT_ENTRIES], hi[HOT_ENTRIES], dw[HOT_ENTRIES], kk[HOT_ENTRIES], wd[HOT_ENTRIES];
#pragma unroll
    for (int e = 0; e < HOT_ENTRIES; ++e) { lo[e] = 1; hi[e] = 0; dw[e] = 0; kk[e] = 0; wd[e] = 0; }
    unsigned prev16 = 0;  // floor(cum(k-1) * 65536), saturated
    // (the block is walked by ONE thread -- up to 18 000 rows: the sums and types of eight rows are requested together, so that a row does not wait for
    // its own loads, and a row's destination is only read if the row is wide enough to enter the list: set_opacity's share of a tardis_example-sized call
    // was mostly this loop, profiles/r06_boundary.txt)
    for (int k0 = b0; k0 < b1; k0 += 8) {
        double cv[8];
        int tv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = k0 + q < b1;
            cv[q] = in ? c[k0 + q] : 0.0;
            tv[q] = in ? ttype[k0 + q] : -2;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            if (k >= b1) break;
            const double t = cv[q] * 65536.0;
            const unsigned cur16 = t >= 65535.0 ? 65535u : (t > 0.0 ? (unsigned)t : 0u);
            const unsigned l = k == b0 ? 0u : prev16 + 1u, h = cur16;
            prev16 = cur16;
            const int tt = tv[q];
            if (h <= l || tt < -1 || k - b0 >= 65535) continue;
            if (h - l <= wd[HOT_ENTRIES - 1]) continue;  // not wider than the narrowest of the six: it would not enter the list
            unsigned d;
            if (tt == -1) d = (unsigned)tline[k] | WALK_EMIT;
            else {
                const int lvl = dest[k];
                d = (unsigned)lvl | ((hot_flag && hot_flag[lvl]) ? WALK_HOT_DEST : 0u);
                if ((unsigned)lvl >= WALK_HOT_DEST) continue;
            }
            unsigned w = h - l, nl = l, nh = h, nd = d, nk = (unsigned)(k - b0);
#pragma unroll
            for (int e = 0; e < HOT_ENTRIES; ++e)  // insertion into the (descending) list of the widest six
                if (w > wd[e]) {
                    unsigned t0 = wd[e]; wd[e] = w; w = t0;
                    t0 = lo[e]; lo[e] = nl; nl = t0;
                    t0 = hi[e]; hi[e] = nh; nh = t0;
                    t0 = dw[e]; dw[e] = nd; nd = t0;
                    t0 = kk[e]; kk[e] = nk; nk = t0;
                }
        }
    }
    unsigned total = 0;
#pragma unroll
    for (int e = 0; e < HOT_ENTRIES; ++e) total += wd[e];
    uint4 *out = reinterpret_cast<uint4 *>(hot_sec + ((size_t)s * (size_t)n_blocks + (size_t)b) * 16);
    out[0] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), hi[0] | (hi[1] << 16));
    out[1] = make_uint4(hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), dw[0], dw[1]);
    out[2] = make_uint4(dw[2], dw[3], dw[4], dw[5]);
    out[3] = make_uint4(kk[0] | (kk[1] << 16), kk[2] | (kk[3] << 16), kk[4] | (kk[5] << 16), total);
    if (mass) mass[(size_t)s * (size_t)n_blocks + (size_t)b] = total;
}

// The second of the two passes: the choice of the blocks that get a hot sector is made (hot_flag); the records of the first pass name their
// destinations by block id -- mark those that are themselves entered through a hot sector.  One thread per (block, shell) touches its 64-byte record
// (what re-running walk_hot_kernel with the flags produced, without walking the blocks again).
__global__ void __launch_bounds__(256) walk_hot_flag_kernel(const unsigned char *__restrict__ hot_flag, long long n_records, unsigned *__restrict__ hot_sec)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_records) return;
    unsigned *rec = hot_sec + (size_t)i * 16;
    const unsigned lo3[3] = {rec[0], rec[1], rec[2]}, hi3[3] = {rec[3], rec[4], rec[5]};
#pragma unroll
    for (int e = 0; e < HOT_ENTRIES; ++e) {
        const unsigned sh16 = 16u * (unsigned)(e & 1);
        const unsigned l = (lo3[e >> 1] >> sh16) & 0xffffu, h = (hi3[e >> 1] >> sh16) & 0xffffu;
        const unsigned d = rec[6 + e];
        if (h > l && !(d & WALK_EMIT) && hot_flag[d & 0x3fffffffu]) rec[6 + e] = d | WALK_HOT_DEST;  // (an empty entry has lo = 1, hi = 0)
    }
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
