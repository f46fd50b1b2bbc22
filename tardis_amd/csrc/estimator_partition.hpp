// estimator_partition.hpp -- the line-estimator passes as a two-level partition of the log RECORDS (round 4, option "est_pipeline" 1).
//
// estimator_log.hpp turns an epoch's log into j_blue / Edotlu by counting-sorting the record INDICES by (shell, 2048-line tile) and
// letting the accumulate kernel fetch its records through that index: a random 4-byte write and a random 24-byte read per record over a
// log of 20 - 60 GB, and the random-access rate of HBM is what bounds both (profiles/r04_estimator_pass_experiments.txt).  Here the records
// themselves are moved, twice, in runs, and never fetched at random:
//
//   bin_count_kernel, bin_scan_kernel (estimator_log.hpp)  -- histogram over (shell, tile) bins and its prefix sums, as before
//   partition_kernel<true>   -- log chunks -> scratch copy, grouped by SHELL
//   partition_kernel<false>  -- scratch copy -> the log's own record buffer, grouped by bin (the input is grouped by shell, so the 2048
//                               records a workgroup stages fall into the ~245 bins of one or two shells)
//   accumulate_blocks_kernel<.., DIRECT>  -- every slice of a bin is a contiguous range of records
// A workgroup stages 2048 records in LDS, ranks them per bucket (one LDS atomic per distinct bucket and wave: the lanes of a wave that
// hold the same bucket find each other with ballots over the key bits), reserves the buckets' next positions with one global atomic
// per non-empty bucket and writes bucket by bucket: neighbouring threads write neighbouring 8-byte words.  The order of the records
// inside a bin is whatever the atomics gave (the estimators are sums: same tolerance as before).
#pragma once
#include "estimator_log.hpp"

namespace mc {

#ifndef TMC_PART_RECORDS
#define TMC_PART_RECORDS 2048
#endif
constexpr int PART_RECORDS = TMC_PART_RECORDS;  // records a workgroup stages at a time (2048: two workgroups per CU; 4096 -- one per CU, twice the run length -- measured
                                                // in round 6: profiles/r06_partition_staging.txt)
constexpr int PART_THREADS = 1024;
constexpr int PART_LOCAL_BUCKETS = 1024;  // buckets a staged segment may span (relative to its first one)

__global__ void __launch_bounds__(256) partition_shell_fill_kernel(const unsigned *__restrict__ bin_start, int tiles_per_shell, int n_shells,
                                                                   unsigned *__restrict__ shell_fill)
{
    for (int sh = blockIdx.x * 256 + threadIdx.x; sh < n_shells; sh += gridDim.x * 256) shell_fill[sh] = bin_start[sh * tiles_per_shell];
}

// MODE 1 (by shell): input = the log's chunks (keys: bin keys; region_count / region_capacity), bucket = shell, bucket_fill[n_shells].
// MODE 0: input = `*total` records in a row, grouped by shell (no keys: the bin follows from the record's first line), bucket = bin,
//         bucket_fill[n_shells * tiles_per_shell].
// MODE 2 (round 6, the shell-sorted log of propagate_wave_kernel<..., SL>): input = the log's chunks, every chunk holding records of ONE shell;
//         bucket = bin (the keys), relative to the first bin of the chunk's shell; bucket_fill[n_shells * tiles_per_shell].
template <int MODE>
__global__ void __launch_bounds__(PART_THREADS) partition_kernel(const LineVisitRecord *__restrict__ records, const unsigned *__restrict__ keys,
                                                                 const unsigned *__restrict__ region_count, int n_regions, unsigned region_capacity,
                                                                 const unsigned *__restrict__ total, int tiles_per_shell, int n_lines, int key_bits,
                                                                 unsigned *__restrict__ bucket_fill, LineVisitRecord *__restrict__ out)
{
    __builtin_amdgcn_s_setprio(3);
    // (records as 8-byte words, 3 per record: loads and stores are coalesced 8-byte accesses)
    __shared__ unsigned long long rec_w[PART_RECORDS * 3];
    __shared__ __attribute__((aligned(16))) unsigned short key_rank[2 * PART_RECORDS];  // local bucket of record i | its rank inside the bucket ...
    unsigned short *key16 = key_rank, *rank16 = key_rank + PART_RECORDS;
    unsigned *dst32 = reinterpret_cast<unsigned *>(key_rank);  // ... and, once every record knows its place, where sorted position j goes in `out`
    __shared__ unsigned short perm[PART_RECORDS];    // record at sorted position j
    __shared__ unsigned hist[PART_LOCAL_BUCKETS], off[PART_LOCAL_BUCKETS], gbase[PART_LOCAL_BUCKETS];
    __shared__ unsigned scratch[PART_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr bool BY_SHELL = MODE == 1, CHUNKS = MODE != 0;
    const unsigned segs_per_region = CHUNKS ? (region_capacity + PART_RECORDS - 1) / PART_RECORDS : 1u;
    const unsigned n_total = CHUNKS ? 0u : *total;
    const unsigned n_segments = CHUNKS ? (unsigned)n_regions * segs_per_region : (n_total + PART_RECORDS - 1) / PART_RECORDS;
    auto bin_of = [&](unsigned idx0) {
        const unsigned shell = idx0 / (unsigned)n_lines, start = idx0 - shell * (unsigned)n_lines;
        return shell * (unsigned)tiles_per_shell + start / (unsigned)EST_TILE;
    };
    for (unsigned seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        size_t base;
        unsigned n;
        if (CHUNKS) {
            const unsigned r = seg / segs_per_region, h = seg - r * segs_per_region;
            const unsigned in_region = min(region_count[r], region_capacity);
            const unsigned first = h * PART_RECORDS;
            n = in_region > first ? min(in_region - first, (unsigned)PART_RECORDS) : 0u;
            base = (size_t)r * region_capacity + first;
        } else {
            base = (size_t)seg * PART_RECORDS;
            n = min(n_total - (unsigned)base, (unsigned)PART_RECORDS);
        }
        if (n == 0) continue;  // (uniform over the workgroup)
        for (int b = tid; b < PART_LOCAL_BUCKETS; b += PART_THREADS) hist[b] = 0;
        {  // (16 bytes per lane and load; an odd record count leaves one word)
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 *__restrict__ src = reinterpret_cast<const u64x2 *>(records + base);
            u64x2 *dst2 = reinterpret_cast<u64x2 *>(rec_w);
            const unsigned n_pairs = (3 * n) >> 1;
            for (unsigned w = tid; w < n_pairs; w += PART_THREADS) dst2[w] = __builtin_nontemporal_load(src + w);
            if ((n & 1u) && tid == 0) rec_w[3 * n - 1] = reinterpret_cast<const unsigned long long *>(records + base)[3 * n - 1];
        }
        __syncthreads();
        // ---- the segment's buckets: [first_bucket, first_bucket + PART_LOCAL_BUCKETS) or, for a segment that spans more (a run of nearly
        // empty shells), the slow way: one global atomic and one scattered write per record
        unsigned first_bucket = 0;
        bool local = true;
        if (MODE == 0) {
            const unsigned sh0 = (unsigned)rec_w[2] / (unsigned)n_lines, sh1 = (unsigned)rec_w[3 * (n - 1) + 2] / (unsigned)n_lines;  // grouped by shell: first, last
            first_bucket = sh0 * (unsigned)tiles_per_shell;
            local = (sh1 - sh0 + 1u) * (unsigned)tiles_per_shell <= (unsigned)PART_LOCAL_BUCKETS;
        }
        if (MODE == 2) first_bucket = (keys[base] / (unsigned)tiles_per_shell) * (unsigned)tiles_per_shell;  // (one shell per chunk; the host checks tiles_per_shell <= PART_LOCAL_BUCKETS)
        if (!local) {
            for (unsigned i = tid; i < n; i += PART_THREADS) {
                const unsigned pos = atomicAdd(&bucket_fill[bin_of((unsigned)rec_w[3 * i + 2])], 1u);
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(out + pos);
                dst[0] = rec_w[3 * i]; dst[1] = rec_w[3 * i + 1]; dst[2] = rec_w[3 * i + 2];
            }
            __syncthreads();
            continue;
        }
        // ---- bucket and rank of every record: the lanes of a wave with the same bucket find each other (ballots over the key bits), the
        // first of them adds their number to the bucket's count
        for (unsigned i0 = (unsigned)(tid & ~63); i0 < n; i0 += PART_THREADS) {  // (wave-uniform trip count)
            const unsigned i = i0 + (unsigned)lane;
            const bool valid = i < n;
            unsigned k = 0;
            if (valid) k = BY_SHELL ? keys[base + i] / (unsigned)tiles_per_shell : (MODE == 2 ? keys[base + i] : bin_of((unsigned)rec_w[3 * i + 2])) - first_bucket;
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < key_bits; ++b) {
                const bool bit = (k >> b) & 1u;
                const unsigned long long m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const int leader = __builtin_ctzll(peers);
                unsigned start = 0;
                if (lane == leader) start = atomicAdd(&hist[k], (unsigned)__popcll(peers));
                start = (unsigned)__shfl((int)start, leader);
                key16[i] = (unsigned short)k;
                rank16[i] = (unsigned short)(start + (unsigned)__popcll(peers & lt_mask));
            }
        }
        __syncthreads();
        // ---- exclusive scan of the histogram (two buckets per thread, then the 8 wave sums) and the reservation of the buckets' output
        // positions: one global atomic per non-empty bucket
        constexpr int PER = PART_LOCAL_BUCKETS / PART_THREADS;
        unsigned cnt[PER], mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { cnt[q] = hist[tid * PER + q]; mine += cnt[q]; }
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) scratch[tid >> 6] = incl;
        __syncthreads();
        unsigned run = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) run += scratch[w];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            off[b] = run;
            if (cnt[q]) gbase[b] = atomicAdd(&bucket_fill[first_bucket + (unsigned)b], cnt[q]);
            run += cnt[q];
        }
        __syncthreads();
        // ---- sorted position of every record and where that position goes (dst32 takes the place of key16 / rank16: read first)
        static_assert(PART_RECORDS % PART_THREADS == 0, "whole records per thread");
        constexpr int PER_T = PART_RECORDS / PART_THREADS;
        unsigned pos[PER_T], where[PER_T];
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const unsigned i = tid + q * PART_THREADS;
            if (i < n) {
                const unsigned k = key16[i], r = rank16[i];
                pos[q] = off[k] + r; where[q] = gbase[k] + r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const unsigned i = tid + q * PART_THREADS;
            if (i < n) { perm[pos[q]] = (unsigned short)i; dst32[pos[q]] = where[q]; }
        }
        __syncthreads();
        // ---- out, word by word: the words of sorted position j go to record dst32[j]; neighbouring threads write neighbouring words
        unsigned long long *__restrict__ dst = reinterpret_cast<unsigned long long *>(out);
        for (unsigned w = tid; w < 3 * n; w += PART_THREADS) {
            const unsigned j = w / 3u, part = w - 3u * j;
            dst[(size_t)dst32[j] * 3 + part] = rec_w[3 * (unsigned)perm[j] + part];
        }
        __syncthreads();
    }
}

}  // namespace mc
