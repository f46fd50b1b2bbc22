// estimator_log.hpp -- deferred accumulation of the line estimators (j_blue, Edotlu).
//
// Why: update_line_estimators (estimators/estimators_line.py) adds to j_blue[line, shell] and Edotlu[line, shell] for
// EVERY line a packet passes, ~375 read-modify-writes of each array per packet in the tardis_example shape.  On MI355X
// device-scope fp64 atomics are executed memory-side (8 XCDs, 8 L2s): the PMC counters show every one of the 1.9e9
// atomic requests of a 1e7-packet launch leaving the L2 (TCC_EA0_ATOMIC == TCC_ATOMIC), and the chip sustains ~2e10 of
// them per second -- that request rate, not arithmetic, bounded the propagation kernel (97 ms; 62 ms with the atomics
// stubbed out).  A trace passes a CONTIGUOUS run of lines of one shell, and the value it adds to each of them is a pure
// function of (line, packet state at the start of the trace) -- in fact a per-trace constant times nu_line (see
// LineVisitRecord in mc_device.hpp).  So the propagation kernel only logs one 24-byte record per trace, and three small
// kernels turn the log into the estimators:
//
//   bin_count_kernel   -- histogram of the records over (shell, TILE-line tile) bins          (LDS histogram)
//   bin_scatter_kernel -- counting sort of the record indices by bin                            (LDS ranks)
//   accumulate_kernel  -- per bin slice: the tile's j_blue/Edotlu live in LDS, every record of the slice adds its two
//                         constants over its range of lines with LDS atomics; the tile is multiplied by nu_line and
//                         added to the global arrays once.
// Round 4: the default is now the pipeline of estimator_partition.hpp (the records themselves are grouped by bin in two passes, then
// accumulate_blocks_kernel below reads them in order); the index pipeline above stays as option "est_pipeline" 0 and for shapes the
// partition kernel's local buckets do not cover.
// The sums agree with the serial reference's to rounding: a different order of the same terms (as with atomics), each term
// within ~1e-14 of the reference's (tests/test_lane_sweep_bounds.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_device.hpp"

namespace mc {

constexpr int EST_TILE = 2048;       // lines per LDS tile (2 x 16 KiB)
constexpr int EST_SLICE = 65536;     // records per accumulate workgroup pass
constexpr int EST_APRON = 256;       // lines past the end of a tile that are still accumulated in LDS (traces start in
                                     // their bin's tile and may run on into the next one)
constexpr int EST_MAX_BINS = 36864;  // largest LDS histogram of the binning kernels (dynamic LDS, 4 B per bin = 144 KiB)

__global__ void __launch_bounds__(256) bin_count_kernel(const unsigned *__restrict__ keys, const unsigned *__restrict__ region_count,
                                                        int n_regions, unsigned region_capacity, int n_bins, unsigned *__restrict__ bin_count)
{
    __builtin_amdgcn_s_setprio(3);  // these passes run beside the next epoch's propagation: its (older) waves would otherwise win every issue slot
    extern __shared__ unsigned hist[];
    for (int b = threadIdx.x; b < n_bins; b += 256) hist[b] = 0;
    __syncthreads();
    for (int r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const unsigned n = min(region_count[r], region_capacity);
        const unsigned *__restrict__ k = keys + (size_t)r * region_capacity;
        for (unsigned i = threadIdx.x; i < n; i += 256) atomicAdd(&hist[k[i]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_bins; b += 256)
        if (hist[b]) atomicAdd(&bin_count[b], hist[b]);
}

// one workgroup: exclusive scan of the bin counts -> bin_start[n_bins + 1]; slices of at most EST_SLICE records ->
// slice_start[n_bins + 1] (prefix of the per-bin slice counts); bin_fill is reset to the bin starts for the scatter.
__global__ void __launch_bounds__(256) bin_scan_kernel(const unsigned *__restrict__ bin_count, int n_bins, unsigned *__restrict__ bin_start,
                                                       unsigned *__restrict__ bin_fill, unsigned *__restrict__ slice_start)
{
    __builtin_amdgcn_s_setprio(3);  // these passes run beside the next epoch's propagation: its (older) waves would otherwise win every issue slot
    auto slices = [&](int b) { return (bin_count[b] + EST_SLICE - 1) / EST_SLICE; };
    __shared__ unsigned part_r[256], part_s[256];
    const int per = (n_bins + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(n_bins, b0 + per);
    unsigned r = 0, s = 0;
    for (int b = b0; b < b1; ++b) { r += bin_count[b]; s += slices(b); }
    part_r[threadIdx.x] = r; part_s[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ar = 0, as = 0;
        for (int k = 0; k < 256; ++k) { const unsigned tr = part_r[k], ts = part_s[k]; part_r[k] = ar; part_s[k] = as; ar += tr; as += ts; }
    }
    __syncthreads();
    r = part_r[threadIdx.x]; s = part_s[threadIdx.x];
    for (int b = b0; b < b1; ++b) {
        bin_start[b] = r; bin_fill[b] = r; slice_start[b] = s;
        r += bin_count[b]; s += slices(b);
    }
    if (b1 == n_bins && b0 <= n_bins) { bin_start[n_bins] = r; slice_start[n_bins] = s; }
}

// counting-sort scatter of the record indices: every workgroup takes a set of log regions, ranks their records per bin
// in LDS, reserves the output ranges with one global atomic per non-empty bin and writes the (global) record indices.
__global__ void __launch_bounds__(256) bin_scatter_kernel(const unsigned *__restrict__ keys, const unsigned *__restrict__ region_count,
                                                          int n_regions, unsigned region_capacity, int n_bins,
                                                          unsigned *__restrict__ bin_fill, unsigned *__restrict__ sorted_index)
{
    __builtin_amdgcn_s_setprio(3);  // these passes run beside the next epoch's propagation: its (older) waves would otherwise win every issue slot
    extern __shared__ unsigned hist[];
    for (int b = threadIdx.x; b < n_bins; b += 256) hist[b] = 0;
    __syncthreads();
    for (int r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const unsigned n = min(region_count[r], region_capacity);
        const unsigned *__restrict__ k = keys + (size_t)r * region_capacity;
        for (unsigned i = threadIdx.x; i < n; i += 256) atomicAdd(&hist[k[i]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_bins; b += 256) {
        const unsigned c = hist[b];
        hist[b] = c ? atomicAdd(&bin_fill[b], c) : 0u;  // base of this block's range in bin b
    }
    __syncthreads();
    for (int r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const unsigned n = min(region_count[r], region_capacity);
        const size_t base = (size_t)r * region_capacity;
        for (unsigned i = threadIdx.x; i < n; i += 256) {
            const unsigned pos = atomicAdd(&hist[keys[base + i]], 1u);
            sorted_index[pos] = (unsigned)(base + i);
        }
    }
}

// Workgroups (8 waves) loop over slices (<= EST_SLICE records of one bin): the bin's tile of both estimators lives in
// LDS.  A wave stages 64 records at a time (the next 128 are already in flight) and spreads their line visits evenly
// over its lanes: lane l of pass i handles visit 64 i + l of the batch; the record a visit belongs to is found with one
// popcount on a bit mask of the record starts.  Records longer than EST_LONG lines are walked by the whole wave.
constexpr int EST_LONG = 63;
constexpr int ACC_WAVES = 8;

template <bool FULL>
__global__ void __launch_bounds__(64 * ACC_WAVES) accumulate_kernel(const LineVisitRecord *__restrict__ records,
                                                                    const unsigned *__restrict__ sorted_index,
                                                                    const unsigned *__restrict__ bin_start,
                                                                    const unsigned *__restrict__ slice_start, int n_bins,
                                                                    int tiles_per_shell, int n_lines, const double *__restrict__ nu_line,
                                                                    double *__restrict__ jblue_t, double *__restrict__ edot_t)
{
    __builtin_amdgcn_s_setprio(3);  // these passes run beside the next epoch's propagation: its (older) waves would otherwise win every issue slot
    constexpr int TILE_LDS = EST_TILE + EST_APRON;
    __shared__ double tile_jb[TILE_LDS], tile_ed[TILE_LDS];
    // staged records, array of structures: a lane fetches "its" record with three 16-byte LDS reads
    struct __attribute__((aligned(8))) Staged { double c_e, c_jb; unsigned idx0, first; };
    __shared__ Staged staged[ACC_WAVES][64];
    __shared__ unsigned long long starts[ACC_WAVES][64];
    const unsigned n_slices = slice_start[n_bins];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    for (unsigned slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        // bin of this slice: last b with slice_start[b] <= slice
        int lo = 0, hi = n_bins;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (slice_start[mid] <= slice) lo = mid; else hi = mid;
        }
        const int bin = lo;
        const unsigned rec_first = bin_start[bin] + (slice - slice_start[bin]) * EST_SLICE;
        const unsigned rec_last = min(bin_start[bin + 1], rec_first + EST_SLICE);
        const int shell = bin / tiles_per_shell, tile = bin - shell * tiles_per_shell;
        const unsigned row = (unsigned)shell * (unsigned)n_lines;
        const unsigned tile_idx0 = row + (unsigned)tile * EST_TILE;
        const unsigned tile_len = min((unsigned)TILE_LDS, (unsigned)n_lines - (unsigned)tile * EST_TILE);  // never past the shell's row
        for (int k = threadIdx.x; k < TILE_LDS; k += 64 * ACC_WAVES) { tile_jb[k] = 0.0; tile_ed[k] = 0.0; }
        __syncthreads();

        // (the tile accumulates the traces' constants; the factor nu_line is applied when the tile is flushed)
        auto add_term = [&](unsigned idx, double c_jb, double c_e) {
            const unsigned off = idx - tile_idx0;
            if (off < tile_len) {
                atomicAdd(&tile_jb[off], c_jb);
                atomicAdd(&tile_ed[off], c_e);
            } else {  // the trace ran far past the end of its first tile
                const double f = FULL ? 1.0 : nu_line[idx - row];
                atomic_add_f64(&jblue_t[idx], c_jb * f);
                atomic_add_f64(&edot_t[idx], c_e * f);
            }
        };
        auto fetch = [&](unsigned r) {
            LineVisitRecord rec;
            rec.c_e = rec.c_jb = 0.0; rec.idx0 = 0; rec.n = 0;
            if (r < rec_last) rec = records[sorted_index[r]];
            return rec;
        };
        const unsigned stride = 64 * ACC_WAVES;
        unsigned base = rec_first + (unsigned)w * 64;
        LineVisitRecord next = fetch(base + (unsigned)lane), next2 = fetch(base + stride + (unsigned)lane);
        for (; base < rec_last; base += stride) {
            const LineVisitRecord rec = next;
            next = next2;
            next2 = fetch(base + 2 * stride + (unsigned)lane);
            const unsigned n_all = rec.n;
            const unsigned n = n_all > (unsigned)EST_LONG ? 0u : n_all;  // long records are handled below
            // exclusive scan of n over the wave
            unsigned incl = n;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, off);
                if (lane >= off) incl += up;
            }
            const unsigned excl = incl - n;
            const unsigned total = (unsigned)__shfl((int)incl, 63);
            const unsigned n_pass = (total + 63) >> 6;  // <= EST_LONG
            starts[w][lane] = 0ull;
            if (n) {  // staged in compacted order: the q-th record that starts is the q-th staged one
                const int pos = __popcll(__ballot(true) & ((1ull << lane) - 1ull));
                Staged st;
                st.c_e = rec.c_e; st.c_jb = rec.c_jb; st.idx0 = rec.idx0; st.first = excl;
                staged[w][pos] = st;
                atomicOr(&starts[w][excl >> 6], 1ull << (excl & 63));
            }
            unsigned rec_base = 0;  // records started before this pass (wave-uniform)
            for (unsigned i = 0; i < n_pass; ++i) {
                const unsigned long long m = starts[w][i];
                const unsigned t = (i << 6) + (unsigned)lane;
                if (t < total) {
                    const unsigned q = rec_base + (unsigned)__popcll(m & le_mask) - 1u;
                    const Staged st = staged[w][q];
                    const unsigned idx = st.idx0 + (t - st.first);
                    add_term(idx, st.c_jb, st.c_e);
                }
                rec_base += (unsigned)__popcll(m);
            }
            // long records: the whole wave walks one record at a time
            unsigned long long longs = __ballot(n_all > (unsigned)EST_LONG);
            while (longs) {
                const int q = __builtin_ctzll(longs);
                longs &= longs - 1;
                const unsigned len = (unsigned)__shfl((int)n_all, q);
                const unsigned idx0 = (unsigned)__shfl((int)rec.idx0, q);
                const double l_ce = __shfl(rec.c_e, q), l_cjb = __shfl(rec.c_jb, q);
                for (unsigned k = lane; k < len; k += 64) add_term(idx0 + k, l_cjb, l_ce);
            }
        }
        __syncthreads();
        for (unsigned k = threadIdx.x; k < tile_len; k += 64 * ACC_WAVES) {
            const double f = FULL ? 1.0 : nu_line[tile_idx0 - row + k];
            if (tile_jb[k] != 0.0) atomic_add_f64(&jblue_t[tile_idx0 + k], tile_jb[k] * f);
            if (tile_ed[k] != 0.0) atomic_add_f64(&edot_t[tile_idx0 + k], tile_ed[k] * f);
        }
        __syncthreads();
    }
}

// accumulate_kernel with BLOCK SUMS (round 4).  On the configs[2] table shape a trace passes 41 lines on average (3300 line visits
// per packet, 80 records): two ds_add_f64 per visit cost 11 LDS cycles per wave instruction each, and ~12 vector instructions per
// record are what a SIMD (one VALU issue per four cycles) has time for.  A record adds the SAME two constants to every line of
// [a, a + n): here the lines up to the next 8-line boundary and those after the last one are added one by one as before, but every
// aligned block of 8 lines in between gets ONE add, into a per-block accumulator behind the tile; when the tile is flushed a line's sum
// is its own accumulator plus its block's.  41 visits become ~11 adds.  Nothing is subtracted anywhere: the terms of a line's sum are
// the reference's, associated differently (as they already are with atomics); a line nobody visited stays exactly 0.
// Measured (profiles/r04_estimator_partition.txt, 1.6e9 records): 35.7 ms reading the records in order (DIRECT), 78 ms through the sorted
// index (bound by the random 24-byte fetches; accumulate_kernel: 85 ms).
constexpr int ACCB_WAVES = 16;                           // two workgroups per CU (72 KB of LDS each)
constexpr int ACCB_TILE = EST_TILE + EST_APRON;          // lines in LDS
constexpr int ACCB_BLOCKS = ACCB_TILE / 8;
constexpr int ACCB_LONG = 255;                           // longer records are walked by the whole wave (<= 7 + 31 + 7 items otherwise)
constexpr int ACCB_PASSES = 48;                          // >= 45 = the items of 64 records of ACCB_LONG lines / 64
static_assert(ACCB_TILE % 8 == 0, "whole blocks");

// DIRECT: the records are grouped by bin already (estimator_partition.hpp): slice r reads records[r], no index.
template <bool FULL, bool DIRECT>
__global__ void __launch_bounds__(64 * ACCB_WAVES, 8) accumulate_blocks_kernel(const LineVisitRecord *__restrict__ records,
                                                                               const unsigned *__restrict__ sorted_index,
                                                                               const unsigned *__restrict__ bin_start,
                                                                               const unsigned *__restrict__ slice_start, int n_bins,
                                                                               int tiles_per_shell, int n_lines, const double *__restrict__ nu_line,
                                                                               double *__restrict__ jblue_t, double *__restrict__ edot_t)
{
    __builtin_amdgcn_s_setprio(3);
    // accumulators: [0, ACCB_TILE) the lines of the tile, [ACCB_TILE, ACCB_TILE + ACCB_BLOCKS) its aligned 8-line blocks
    __shared__ double acc_jb[ACCB_TILE + ACCB_BLOCKS], acc_ed[ACCB_TILE + ACCB_BLOCKS];
    // staged records: the constants; a | h << 12 | b << 15 (first line relative to the tile, lines before the first boundary, whole
    // blocks); the record's first item in the batch
    struct __attribute__((aligned(8))) Staged { double c_e, c_jb; unsigned ahb, first; };
    __shared__ Staged staged[ACCB_WAVES][64];
    __shared__ unsigned long long starts[ACCB_WAVES][ACCB_PASSES];
    const unsigned n_slices = slice_start[n_bins];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // bits of a wave mask below this lane, counted without holding the lane's own mask in registers
    auto count_below = [](unsigned long long m) {
        return (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };
    for (unsigned slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        int lo = 0, hi = n_bins;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (slice_start[mid] <= slice) lo = mid; else hi = mid;
        }
        const int bin = lo;
        const unsigned rec_first = bin_start[bin] + (slice - slice_start[bin]) * EST_SLICE;
        const unsigned rec_last = min(bin_start[bin + 1], rec_first + EST_SLICE);
        const int shell = bin / tiles_per_shell, tile = bin - shell * tiles_per_shell;
        const unsigned row = (unsigned)shell * (unsigned)n_lines;
        const unsigned tile_idx0 = row + (unsigned)tile * EST_TILE;
        const unsigned tile_len = min((unsigned)ACCB_TILE, (unsigned)n_lines - (unsigned)tile * EST_TILE);  // never past the shell's row
        for (int k = threadIdx.x; k < ACCB_TILE + ACCB_BLOCKS; k += 64 * ACCB_WAVES) { acc_jb[k] = 0.0; acc_ed[k] = 0.0; }
        __syncthreads();
        auto far_line = [&](unsigned o, double c_jb, double c_e) {  // a line past the LDS tile (o: relative to the tile)
            const double f = FULL ? 1.0 : nu_line[tile_idx0 - row + o];
            atomic_add_f64(&jblue_t[tile_idx0 + o], c_jb * f);
            atomic_add_f64(&edot_t[tile_idx0 + o], c_e * f);
        };
        // item j of a record {a, h, b}: the h lines before its first 8-line boundary, its b whole blocks, the lines after the last boundary
        // (a record never runs past its shell's row, so a block it covers exists in full)
        auto add_item = [&](unsigned a, unsigned h, unsigned b, unsigned j, double c_jb, double c_e) {
            const unsigned jh = j - h;                                  // (wraps for j < h: not a block then)
            const bool is_blk = jh < b;
            const unsigned line = a + j + (j >= h + b ? 7u * b : 0u);   // after the blocks: a + h + 8 b + (j - h - b)
            const unsigned blk_line = a + h + 8u * jh;                  // first line of block jh
            const unsigned idx = is_blk ? (unsigned)ACCB_TILE + (blk_line >> 3) : line;
            const unsigned last = is_blk ? blk_line + 7u : line;        // the last line the item covers
            if (last < tile_len) {
                atomicAdd(&acc_jb[idx], c_jb);
                atomicAdd(&acc_ed[idx], c_e);
            } else if (is_blk) {
                for (unsigned k = 0; k < 8u; ++k) {
                    if (blk_line + k < tile_len) { atomicAdd(&acc_jb[blk_line + k], c_jb); atomicAdd(&acc_ed[blk_line + k], c_e); }
                    else far_line(blk_line + k, c_jb, c_e);
                }
            } else far_line(line, c_jb, c_e);
        };
        auto fetch = [&](unsigned r) {
            LineVisitRecord rec;
            rec.c_e = rec.c_jb = 0.0; rec.idx0 = tile_idx0; rec.n = 0;
            if (r < rec_last) rec = records[DIRECT ? r : sorted_index[r]];
            return rec;
        };
        const unsigned stride = 64 * ACCB_WAVES;
        unsigned base = rec_first + (unsigned)w * 64;
        LineVisitRecord next = fetch(base + (unsigned)lane);  // (32 waves per CU: one batch ahead is enough)
        for (; base < rec_last; base += stride) {
            const LineVisitRecord rec = next;
            next = fetch(base + stride + (unsigned)lane);
            const unsigned n_all = rec.n, a = rec.idx0 - tile_idx0;  // a < EST_TILE
            const unsigned h = min(n_all, (8u - (a & 7u)) & 7u), b = (n_all - h) >> 3, tl = (n_all - h) & 7u;
            const unsigned m = n_all > (unsigned)ACCB_LONG ? 0u : h + b + tl;  // (<= 45 items; long records are handled below)
            unsigned incl = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, off);
                if (lane >= off) incl += up;
            }
            const unsigned excl = incl - m;
            const unsigned total = (unsigned)__shfl((int)incl, 63);
            const unsigned n_pass = (total + 63) >> 6;  // <= 45
            if (lane < ACCB_PASSES) starts[w][lane] = 0ull;
            if (m) {  // staged in compacted order: the q-th record that starts is the q-th staged one
                const unsigned pos = count_below(__ballot(true));
                Staged st;
                st.c_e = rec.c_e; st.c_jb = rec.c_jb; st.ahb = a | (h << 12) | (b << 15); st.first = excl;
                staged[w][pos] = st;
                atomicOr(&starts[w][excl >> 6], 1ull << (excl & 63));
            }
            unsigned rec_base = 0;  // records started before this pass (wave-uniform)
            for (unsigned i = 0; i < n_pass; ++i) {
                const unsigned long long mk = starts[w][i];
                const unsigned t = (i << 6) + (unsigned)lane;
                if (t < total) {
                    // the record of item t: the last one that starts at or before it
                    const unsigned q = rec_base + count_below(mk) + (unsigned)((mk >> lane) & 1ull) - 1u;
                    const Staged st = staged[w][q];
                    add_item(st.ahb & 0xfffu, (st.ahb >> 12) & 7u, st.ahb >> 15, t - st.first, st.c_jb, st.c_e);
                }
                rec_base += (unsigned)__popcll(mk);
            }
            // long records: the whole wave walks the items of one record at a time
            unsigned long long longs = __ballot(n_all > (unsigned)ACCB_LONG);
            while (longs) {
                const int q = __builtin_ctzll(longs);
                longs &= longs - 1;
                const unsigned q_a = (unsigned)__shfl((int)a, q), q_h = (unsigned)__shfl((int)h, q), q_b = (unsigned)__shfl((int)b, q);
                const unsigned q_m = q_h + q_b + (unsigned)__shfl((int)tl, q);
                const double l_ce = __shfl(rec.c_e, q), l_cjb = __shfl(rec.c_jb, q);
                for (unsigned k = lane; k < q_m; k += 64) add_item(q_a, q_h, q_b, k, l_cjb, l_ce);
            }
        }
        __syncthreads();
        for (unsigned k = threadIdx.x; k < tile_len; k += 64 * ACCB_WAVES) {
            const double f = FULL ? 1.0 : nu_line[tile_idx0 - row + k];
            const double v_jb = acc_jb[k] + acc_jb[ACCB_TILE + (k >> 3)], v_ed = acc_ed[k] + acc_ed[ACCB_TILE + (k >> 3)];
            if (v_jb != 0.0) atomic_add_f64(&jblue_t[tile_idx0 + k], v_jb * f);
            if (v_ed != 0.0) atomic_add_f64(&edot_t[tile_idx0 + k], v_ed * f);
        }
        __syncthreads();
    }
}

// accumulate_blocks_kernel with a DYADIC hierarchy of block sums (round 6, option "est_accumulate" 2).  The two-level form above turns a trace of
// 41 lines into ~11 LDS adds per estimator and is bound by them (ds_add_f64: 11 LDS cycles per wave instruction, serialised further by
// bank conflicts between the items of different records; LDS 79 % busy, profiles/r04_estimator_partition.txt).  Here the accumulators are a
// segment tree over the tile -- levels of 1, 2, 4, 8, 16 and 32 lines -- and a record [a, e) adds its two constants to the minimal set of
// aligned dyadic blocks that tile it: with M the highest-level boundary inside (a, e] (the bits of a and e agree above their highest differing
// bit d, M = e with the bits below d cleared), the blocks left of M are the set bits of up = M - a from the lowest up, those right of it the
// set bits of dn = e - M; bits above the top level count as that many 32-line blocks.  41 lines -> ~5.3 adds per estimator.  A line's sum at
// the flush is the sum of the six accumulators above it.  As before nothing is subtracted anywhere (the terms of a line's sum are the
// reference's, associated differently; a line nobody visited stays exactly 0) and the per-term arithmetic is unchanged.
constexpr int ACCD_WAVES = 16;                                 // one workgroup per CU (100 KB of LDS)
constexpr int ACCD_TILE = EST_TILE + EST_APRON;                // lines in LDS
constexpr int ACCD_TOP = 5;                                    // top level: blocks of 32 lines
constexpr int ACCD_CELLS = 2 * ACCD_TILE;                     // accumulators per estimator, laid out like a bottom-up segment tree: the block of 2^l lines at line p is cell (T + p) >> l
                                                               // (T a multiple of 2^TOP: level l fills [T >> l, 2 T >> l), the levels do not overlap; cells below T >> TOP are unused)
constexpr int ACCD_LONG = 255;                                 // longer records are walked by the whole wave
constexpr int ACCD_PASSES = 20;                                // >= 18 = the items of 64 records of ACCD_LONG lines (<= 5 + 8 + 5 each) / 64
static_assert(ACCD_TILE % 256 == 0 && ACCD_TILE % (1 << ACCD_TOP) == 0 && EST_TILE % (1 << ACCD_TOP) == 0, "whole top-level blocks, tiles aligned to them");

// LOOP: every lane walks its own record from left to right -- at line p the largest aligned block that still fits, min(ctz(p | 32), floor(log2(end - p))): the same
// blocks as the item list of the balanced form (ascending towards the split point, 32-line blocks across it, descending behind it) without the list: no scan over
// the item counts, no staging, no item decoding -- ~20 instructions per block instead of ~100 per 64 items + ~110 per batch -- at the price of lanes idling while
// the record with the most blocks in the wave finishes.
template <bool FULL, bool DIRECT, bool LOOP = false>
__global__ void __launch_bounds__(64 * ACCD_WAVES) accumulate_dyadic_kernel(const LineVisitRecord *__restrict__ records,
                                                                            const unsigned *__restrict__ sorted_index,
                                                                            const unsigned *__restrict__ bin_start,
                                                                            const unsigned *__restrict__ slice_start, int n_bins,
                                                                            int tiles_per_shell, int n_lines, const double *__restrict__ nu_line,
                                                                            double *__restrict__ jblue_t, double *__restrict__ edot_t)
{
    __builtin_amdgcn_s_setprio(3);
    __shared__ double acc_jb[ACCD_CELLS], acc_ed[ACCD_CELLS];
    // staged records: the constants; a | up << 11 | dn << 19 (first line relative to the tile, lines left / right of the record's split point); the
    // record's first item in the batch
    struct __attribute__((aligned(8))) Staged { double c_e, c_jb; unsigned aud, first; };
    __shared__ Staged staged[ACCD_WAVES][64];
    __shared__ unsigned long long starts[ACCD_WAVES][ACCD_PASSES];
    __shared__ unsigned short nth_bit[32];  // nth_bit[v] >> 3 j & 7: position of the j-th lowest set bit of the 5-bit value v
    if (threadIdx.x < 32) {
        unsigned t = 0, k = 0;
        for (unsigned b = 0; b < 5; ++b)
            if ((threadIdx.x >> b) & 1u) { t |= b << (3 * k); ++k; }
        nth_bit[threadIdx.x] = (unsigned short)t;
    }
    const unsigned n_slices = slice_start[n_bins];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    auto count_below = [](unsigned long long m) {
        return (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    };
    // split of a record [a, a + n), n >= 1: up lines left of M, dn right of it (see above)
    auto split = [](unsigned a, unsigned n, unsigned &up, unsigned &dn) {
        const unsigned e = a + n;
        const unsigned d = 31u - (unsigned)__builtin_clz(a ^ e);  // (a != e)
        const unsigned M = (e >> d) << d;
        up = M - a; dn = e - M;
    };
    auto n_items = [](unsigned up, unsigned dn) {
        return (unsigned)__builtin_popcount(up & 31u) + (up >> 5) + (dn >> 5) + (unsigned)__builtin_popcount(dn & 31u);
    };
    for (unsigned slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        int lo = 0, hi = n_bins;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (slice_start[mid] <= slice) lo = mid; else hi = mid;
        }
        const int bin = lo;
        const unsigned rec_first = bin_start[bin] + (slice - slice_start[bin]) * EST_SLICE;
        const unsigned rec_last = min(bin_start[bin + 1], rec_first + EST_SLICE);
        const int shell = bin / tiles_per_shell, tile = bin - shell * tiles_per_shell;
        const unsigned row = (unsigned)shell * (unsigned)n_lines;
        const unsigned tile_idx0 = row + (unsigned)tile * EST_TILE;
        const unsigned tile_len = min((unsigned)ACCD_TILE, (unsigned)n_lines - (unsigned)tile * EST_TILE);  // never past the shell's row
        for (int k = threadIdx.x; k < ACCD_CELLS; k += 64 * ACCD_WAVES) { acc_jb[k] = 0.0; acc_ed[k] = 0.0; }
        __syncthreads();
        auto far_line = [&](unsigned o, double c_jb, double c_e) {  // a line past the LDS tile (o: relative to the tile)
            const double f = FULL ? 1.0 : nu_line[tile_idx0 - row + o];
            atomic_add_f64(&jblue_t[tile_idx0 + o], c_jb * f);
            atomic_add_f64(&edot_t[tile_idx0 + o], c_e * f);
        };
        // item j of a record {a, up, dn}: the set bits of up & 31 (blocks growing towards M), the 32-line blocks across M, the set bits of dn & 31
        auto add_item = [&](unsigned a, unsigned up, unsigned dn, unsigned j, double c_jb, double c_e) {
            const unsigned u5 = up & 31u, d5 = dn & 31u;
            const unsigned pu = (unsigned)__builtin_popcount(u5), mid = (up >> 5) + (dn >> 5);
            unsigned level, pos;
            if (j < pu) {
                level = ((unsigned)nth_bit[u5] >> (3u * j)) & 7u;
                pos = a + (u5 & ((1u << level) - 1u));
            } else if (j - pu < mid) {
                level = (unsigned)ACCD_TOP;
                pos = a + u5 + ((j - pu) << ACCD_TOP);
            } else {
                level = ((unsigned)nth_bit[d5] >> (3u * (j - pu - mid))) & 7u;
                pos = a + up + dn - (d5 & ((2u << level) - 1u));
            }
            const unsigned last = pos + (1u << level) - 1u;  // the last line the item covers
            if (last < tile_len) {
                const unsigned idx = ((unsigned)ACCD_TILE + pos) >> level;
                atomicAdd(&acc_jb[idx], c_jb);
                atomicAdd(&acc_ed[idx], c_e);
            } else {
                for (unsigned k = pos; k <= last; ++k) {
                    if (k < tile_len) { atomicAdd(&acc_jb[ACCD_TILE + k], c_jb); atomicAdd(&acc_ed[ACCD_TILE + k], c_e); }
                    else far_line(k, c_jb, c_e);
                }
            }
        };
        // (an unconditional load from a clamped index: behind a branch the loads of the batches ahead cannot be counted and the wait for THIS batch's record
        // becomes a wait for all of them -- vmcnt(0) with the prefetches just issued)
        auto fetch = [&](unsigned r) {
            const unsigned rc = min(r, rec_last - 1u);  // (rec_first < rec_last: a slice is never empty)
            LineVisitRecord rec = records[DIRECT ? rc : sorted_index[rc]];
            if (r >= rec_last) { rec.n = 0; rec.idx0 = tile_idx0; }
            return rec;
        };
        const unsigned stride = 64 * ACCD_WAVES;
        unsigned base = rec_first + (unsigned)w * 64;
        // a record of more than ACCD_LONG lines: the whole wave walks its items (it may leave the LDS tile: add_item checks)
        auto long_records = [&](const LineVisitRecord &rec) {
            unsigned long long longs = __ballot(rec.n > (unsigned)ACCD_LONG);
            while (longs) {
                const int q = __builtin_ctzll(longs);
                longs &= longs - 1;
                const unsigned q_a = (unsigned)__shfl((int)(rec.idx0 - tile_idx0), q), q_n = (unsigned)__shfl((int)rec.n, q);
                unsigned q_up, q_dn;
                split(q_a, q_n, q_up, q_dn);
                const unsigned q_m = n_items(q_up, q_dn);
                const double l_ce = __shfl(rec.c_e, q), l_cjb = __shfl(rec.c_jb, q);
                for (unsigned k = lane; k < q_m; k += 64) add_item(q_a, q_up, q_dn, k, l_cjb, l_ce);
            }
        };
        if (LOOP) {
            // (records of <= ACCD_LONG lines end inside the LDS tile and inside the shell's row: no bounds to check.  Two records per lane walked in one loop --
            // the sum of two records' blocks varies less over the lanes than one record's -- was tried: the switch from the first to the second costs every
            // iteration what the better balance gains.)
            LineVisitRecord next = fetch(base + (unsigned)lane), next2 = fetch(base + stride + (unsigned)lane);
            for (; base < rec_last; base += stride) {
                const LineVisitRecord rec = next;
                next = next2;
                next2 = fetch(base + 2 * stride + (unsigned)lane);
                // (q = T + line: T is a multiple of 256, so q's low bits are the line's, and the cell of a block is q >> level)
                unsigned q = (unsigned)ACCD_TILE + (rec.idx0 - tile_idx0);
                const unsigned e = rec.n > (unsigned)ACCD_LONG ? q : q + rec.n;
                const double cj = rec.c_jb, ce = rec.c_e;
                while (q < e) {
                    const unsigned fit = 31u - (unsigned)__builtin_clz(e - q);
                    const unsigned lv = min((unsigned)__builtin_ctz(q | (1u << ACCD_TOP)), fit);
                    const unsigned idx = q >> lv;
                    atomicAdd(&acc_jb[idx], cj);
                    atomicAdd(&acc_ed[idx], ce);
                    q += 1u << lv;
                }
                long_records(rec);
            }
        } else {
        LineVisitRecord next = fetch(base + (unsigned)lane), next2 = fetch(base + stride + (unsigned)lane);  // (16 waves per CU: two batches ahead)
        for (; base < rec_last; base += stride) {
            const LineVisitRecord rec = next;
            next = next2;
            next2 = fetch(base + 2 * stride + (unsigned)lane);
            const unsigned n_all = rec.n, a = rec.idx0 - tile_idx0;  // a < EST_TILE
            unsigned up = 0, dn = 0;
            if (n_all) split(a, n_all, up, dn);
            const unsigned m = n_all > (unsigned)ACCD_LONG ? 0u : n_items(up, dn);  // (<= 18 items; long records are handled below)
            unsigned incl = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned upv = (unsigned)__shfl_up((int)incl, off);
                if (lane >= off) incl += upv;
            }
            const unsigned excl = incl - m;
            const unsigned total = (unsigned)__shfl((int)incl, 63);
            const unsigned n_pass = (total + 63) >> 6;
            if (lane < ACCD_PASSES) starts[w][lane] = 0ull;
            if (m) {  // staged in compacted order: the q-th record that starts is the q-th staged one
                const unsigned pos = count_below(__ballot(true));
                Staged st;
                st.c_e = rec.c_e; st.c_jb = rec.c_jb; st.aud = a | (up << 11) | (dn << 19); st.first = excl;
                staged[w][pos] = st;
                atomicOr(&starts[w][excl >> 6], 1ull << (excl & 63));
            }
            unsigned rec_base = 0;  // records started before this pass (wave-uniform)
            for (unsigned i = 0; i < n_pass; ++i) {
                const unsigned long long mk = starts[w][i];
                const unsigned t = (i << 6) + (unsigned)lane;
                if (t < total) {
                    // the record of item t: the last one that starts at or before it
                    const unsigned q = rec_base + count_below(mk) + (unsigned)((mk >> lane) & 1ull) - 1u;
                    const Staged st = staged[w][q];
                    add_item(st.aud & 0x7ffu, (st.aud >> 11) & 0xffu, st.aud >> 19, t - st.first, st.c_jb, st.c_e);
                }
                rec_base += (unsigned)__popcll(mk);
            }
            // long records: the whole wave walks the items of one record at a time
            unsigned long long longs = __ballot(n_all > (unsigned)ACCD_LONG);
            while (longs) {
                const int q = __builtin_ctzll(longs);
                longs &= longs - 1;
                const unsigned q_a = (unsigned)__shfl((int)a, q), q_up = (unsigned)__shfl((int)up, q), q_dn = (unsigned)__shfl((int)dn, q);
                const unsigned q_m = n_items(q_up, q_dn);
                const double l_ce = __shfl(rec.c_e, q), l_cjb = __shfl(rec.c_jb, q);
                for (unsigned k = lane; k < q_m; k += 64) add_item(q_a, q_up, q_dn, k, l_cjb, l_ce);
            }
        }
        }
        __syncthreads();
        for (unsigned k = threadIdx.x; k < tile_len; k += 64 * ACCD_WAVES) {
            const double f = FULL ? 1.0 : nu_line[tile_idx0 - row + k];
            double v_jb = acc_jb[ACCD_TILE + k], v_ed = acc_ed[ACCD_TILE + k];
#pragma unroll
            for (unsigned l = 1; l <= (unsigned)ACCD_TOP; ++l) {
                const unsigned idx = ((unsigned)ACCD_TILE + k) >> l;
                v_jb += acc_jb[idx]; v_ed += acc_ed[idx];
            }
            if (v_jb != 0.0) atomic_add_f64(&jblue_t[tile_idx0 + k], v_jb * f);
            if (v_ed != 0.0) atomic_add_f64(&edot_t[tile_idx0 + k], v_ed * f);
        }
        __syncthreads();
    }
}

}  // namespace mc
