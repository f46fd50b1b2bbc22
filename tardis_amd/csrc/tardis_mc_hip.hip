// tardis_mc_hip.hip -- host side of libtardis_mc_hip.so: the C ABI of include/tardis_mc.h over the HIP kernels.
//
// One context = one HIP device + one stream.  Inputs are copied to HBM and re-laid shell-major once
// (tardis_mc_set_*), kernels run asynchronously on the context stream (tardis_mc_propagate), results are
// re-laid to the reference's [L,S] layout on the device and copied out (tardis_mc_get_results).
// RCCL is bound lazily with dlopen so that single-GPU users never map librccl.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <numeric>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tardis_mc.h"
#include "mc_device.hpp"
#include "propagate_lane.hpp"
#include "propagate_group.hpp"
#include "estimator_log.hpp"
#include "estimator_partition.hpp"
#include "propagate_wave.hpp"
#include "packet_source.hpp"
#include "formal_integral.hpp"
#include "tau_prefix.hpp"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes ? bytes : 16;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

// ---- RCCL, bound lazily
struct Id128 { char bytes[TARDIS_MC_UNIQUE_ID_BYTES]; };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ Id128, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

bool load_rccl(std::string &err)
{
    if (g_rccl.handle) return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { err = std::string("dlopen(librccl.so) failed: ") + dlerror(); return false; }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        err = "librccl.so lacks an expected nccl* symbol";
        dlclose(h);
        return false;
    }
    g_rccl.handle = h;
    return true;
}

}  // namespace

struct TardisMcContext {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> ev_chunk;  // 3 per chunk of the cooperative path: before seed / after seed / after propagate
    int chunks_timed = 0;
    bool timed = false;
    std::string err;
    hipDeviceProp_t prop{};
    // state flags
    bool have_geometry = false, have_opacity = false, have_config = false, have_packets = false;
    // geometry
    int n_shells = 0;
    double t_exp = 0;
    DevBuf r_inner, r_outer;
    // opacity
    int n_lines = 0, n_trans = 0, n_levels = 0;
    DevBuf nu_line, tau_t, n_e, prob_t, cum_t, trans_nu, line2level, block_edge, ttype, dest, tline, staging, line_block, trans_rec, bucket_first;
    DevBuf cum16, rec16, quad_info, line_block_c;  // compact walk tables (walk_tables.hpp)
    DevBuf hot_sec, hot_mass, hot_flag, blk_tab;   // hot sectors of the macro-atom walk (walk_tables.hpp)
    bool have_hot = false;
    long long n_hot_blocks = 0;                    // blocks entered through a hot sector (diagnostic)
    int walk_hot = -1;                             // -1: blocks whose six widest intervals cover enough (walk_hot_min_mass); 0: none; 1: every block
    // per mille of a block's probability, mean over the shells; blocks of <= 32 / more rows.  Measured (profiles/r04_walk_hot_sectors.txt):
    // heavy-tailed blocks -11 % whatever the thresholds (95 % of their jumps are decided by the sector); blocks of 12-24 rows with
    // uniformly drawn probabilities (six intervals cover ~50-70 %) lose 2.5 % at 600 -- a missed probe costs a round of the walk
    int walk_hot_min_mass = 800, walk_hot_min_mass_long = 400;
    DevBuf tau_pfx, tau_rowsum, pfx_flag;          // v-packet screening tables (tau_prefix.hpp), built at the first SCREENING call after set_opacity (+ their negative-depth flag)
    bool pfx_valid = false, pfx_negative = false;
    unsigned cum16_stride = 0;
    bool have_walk_tables = false;
    int bucket_shift = 0, bucket_n = 0;
    long long vpk_wave_min_packets = 100000;  // option: calls with v-packets on fine grids take the wave kernel from this many packets on (the group kernel below)
    // Two instantiations of the lane-sweep kernel: 128 VGPRs / sixteen waves per CU / eight lines per step (A), and 166 VGPRs / twelve waves per CU /
    // twelve lines per step (B).  A wins where the steady state dominates (1e8 packets of the configs[2] shape: B +3.8 %), B where the drain of a
    // call's longest packets does (1.25e7 packets: -11 %; 1e5-1e6 packets of the tardis_example shape: -6 %; profiles/r05_ls_instantiations.txt):
    // fewer steps per trace shorten the serial chain of a lone packet, and a mostly idle chip does not miss the four waves.  Option
    // ls_waves_per_simd: 4 = A, 3 = B, 0 (default) = the engine times both on the first calls of a given (packet count, tables) and keeps the
    // faster one -- a Monte Carlo iteration repeats the same call; per-packet results are bit-identical either way.
    int ls_waves_per_simd = 0;
    // Interleaved sweep table (round 6): nt_t[shell][line] = {nu_line, tau}, 16 bytes per line, rows on 128-byte boundaries -- the eight 16-byte loads of
    // a lane-sweep step come from one run of 128 bytes instead of two runs of 64 bytes in two tables.  Option sweep_table: 0 the separate tables,
    // 1 runs from the current line, 2 aligned runs (propagate_wave_kernel<..., NT>); built by the first propagate call after set_opacity that uses it.
    // -1 (default) = 1 -- measured (profiles/r06_sweep_table.txt): sixteen-wave instantiation, 1e8 packets of configs[2] -1.0 ... -1.5 %, 2e7 -4 %, 1.25e7 -6 %,
    // uniform levels -4.5 ... -6 %, configs[1] -1 ... -3 %; twelve-wave instantiation (with the lean proof) -1.7 ... -4.6 %; aligned runs (2) +2 ... +6 % on the
    // heavy-tailed tables (a trace's first step is shorter: 8 % more steps), -5 % on the uniform ones.
    int sweep_table = -1;
    // Shell-sorted log (round 6; propagate_wave_kernel<..., SL>): every chunk of the line-visit log holds records of one shell, the estimator passes start with
    // the partition by bin.  1 / -1: where the kernel has it (the production lane-sweep instantiations, <= 64 shells, partition pipeline); 0 (default) off.
    // Measured (profiles/r06_shell_sorted_log.txt): the passes of an epoch of 2e9 records 86.5 -> 64.2 ms as priced -- and the propagation launches +4.3 %
    // (2773 -> 2891-2904 ms per 1e8 packets) whether the slots come from ballot ranks (+3 % instructions) or from one LDS atomic (+1.5 %): net +0.6 ... +1.3 % on
    // the headline, -2 ... -3 % only where a call's passes are not overlapped by anything (2e7 packets with log_sets 1; configs[1])
    int log_by_shell = 0;
    // Split launches (round 6): from the second epoch of a call on, the propagation grid is launched as TWO kernels -- the first half of the waves at once, on the
    // engine's stream; the second half on the passes' stream, behind the estimator passes of the previous epoch.  While those passes run, the chip holds eight
    // propagation waves per CU instead of none (the passes' 1024-thread workgroups need half a CU: they cannot be placed beside sixteen resident waves, and used to
    // run alone at every epoch boundary: 0.37 s of a 3.1-s step); when they are over the second half follows.  Nothing in the kernel changes: the second launch
    // gets a WaveCold whose per-wave pointers (MT19937 states, suspended lanes / waves, v-packet scratch) are offset by the first launch's wave count.
    // MEASURED, and a clear loss (profiles/r06_split_launch.txt): parity-green, but configs[2] at 1e8 packets 3 109 -> 4 107 ms -- the passes beside eight resident
    // waves per CU take three times as long as alone (their workgroups need 72 KB of CONTIGUOUS LDS and find it on few CUs), the second half of the grid idles
    // meanwhile.  Option epoch_split, default 0.
    int epoch_split = 0;
    hipEvent_t ev_split[3] = {nullptr, nullptr, nullptr};  // the first launch's inputs are in place | start / end of the second launch
    DevBuf nt_t;
    bool nt_valid = false, nt_negative = false;  // (nt_negative: the table holds a negative / NaN optical depth -- the lean proof of the NT kernels does not apply)
    unsigned nt_stride = 0;
    // (the tuner: calls 0-4 of a key run A untimed, A, B, A, B -- each timed call with the tuner's OWN event pair, recorded only when the call was
    // enqueued completely, so that neither another entry point's use of ev_start / ev_stop nor a failed call can leave a stale or unpaired
    // measurement behind -- and from call 5 on B only if the faster of its two calls beat the faster of A's by >= 3 %: boxes differ by more
    // than one noisy sample can tell apart)
    struct { long long n = -1; int lines = 0, shells = 0, mode = 0, table = 0, phase = 0, pending = -1, choice = 0; double ms[2][2] = {{-1.0, -1.0}, {-1.0, -1.0}}; } ls_tune;
    hipEvent_t ev_tune[2] = {nullptr, nullptr};
    int vpk_wide_registers = 1;       // option: the two-waves-per-SIMD v-packet instantiation where LDS bounds the occupancy at eight waves per CU anyway
    int bucket_lines_permille = 750;  // option: target lines per bucket x 1000 (takes effect in set_opacity)
    long long bucket_kmin = 0;
    bool lines_sorted = true;  // line_list_nu strictly usable by the index-based kernels (non-increasing, positive)
    // estimators: one allocation [J | nubar | vhist | pad | jblue copy0 | edot copy0 | jblue copy1.. | edot copy1..]
    DevBuf est;
    size_t est_S = 0, est_L = 0, est_G = 0;
    int est_copies = 1;
    bool est_valid = false;
    // config
    TardisMcConfig cfg{};
    std::vector<double> grid_host;
    DevBuf grid;
    // packets
    long long n_packets = 0;
    DevBuf r0, mu0, nu0, e0, seeds, out_nu, out_e;
    DevBuf li_f64[9], li_i64[5], li_rec;  // (li_rec: the wave kernel's 64-byte tracker records, unpacked into the arrays after the propagation)
    bool track = true;
    // v-packet log
    DevBuf vlog_count, vlog_packet, vlog_seq, vlog_nu, vlog_energy, vlog_mu, vlog_r;
    long long vlog_capacity = 0;
    bool vlog_capacity_user = false;  // set through the vpacket_log_capacity option (otherwise sized per propagate call)
    // scratch
    DevBuf rng_state, counters, first_error, next_packet, seeded_states, problem_dev;
    mc::DeviceProblem problem_host{};
    long long chunk_packets = 16LL << 20;  // packets per seeded-state chunk (2496 B each) of the cooperative kernel
    // launch geometry
    int variant = -1;  // 0: lane-per-packet kernel; 1: group-per-packet kernel; 2: wave-owner kernel, group sweeps; 3: wave-owner kernel, lane sweeps; -1: automatic
    bool prob_negative = false;  // a negative transition probability: the running sums are not monotone, no jump search
    bool walk_min_active_user = false, ls_min_active_user = false;  // (set through the options: no automatic choice then)
    int walk_min_active = 8;  // compact macro-atom walk: carry the longest chains over to the next pass once this few lanes still walk (-1: never)
    int ls_min_active = 8, ls_max_steps = 1 << 30;  // lane sweeps: when to leave the sweep phase (propagate_wave.hpp)
    int blocks_per_cu = 16;
    int debug_flags = 0;
    int drain_split = 0;          // wave kernel: split the drain of a call off into a launch of its own (WaveCold::drain_split); measured: -4 % at 1e7 packets, +1.3 % at 1e8 -> off
    int walk_sector_packing = 1;  // compact walk tables: short blocks do not straddle 64-byte sectors (set before set_opacity; 0: packed at 16 bytes as in round 2)
    int vpacket_screening = -1;  // v-packet screening on the prefix sums of tau (tau_prefix.hpp): -1 automatic, 0 off, 1 on
    int group_size = 0;  // 0: automatic (8 or 16 lanes per packet)
    int log_tail_packets = 8;               // tail split: packets' worth of traces a lane in flight still logs after the supply has run out
    int log_tail_split = 1;                 // plan the epochs so that the last one holds only the drain of the call (see tardis_mc_propagate)
    int est_accumulate = 3;                 // accumulate kernel: 3 dyadic hierarchy of block sums, a lane per record (accumulate_dyadic_kernel<.., LOOP>), 2 the same with the blocks of 64 records spread over the lanes, 1 8-line block sums (accumulate_blocks_kernel), 0 one add per line visit (accumulate_kernel, index pipeline only)
    int est_pipeline = 1;                   // line-estimator passes: 1 two-level partition of the records (estimator_partition.hpp), 0 index sort + gather (estimator_log.hpp)
    DevBuf log_part;                        // est_pipeline 1: the scratch copy of one epoch's records, shared by both buffer sets
    long long log_chunk_records = 0;        // records per chunk of the line-visit log's pool (0: automatic, <= 4096; tests)
    long long log_capacity = 2500000000LL;  // upper bound of the line-visit records per epoch and buffer set of the wave kernel (24 B + 4 B + 4 B each)
    bool log_capacity_user = false;         // set through the log_capacity option (otherwise also bounded by the free device memory)
    // wave kernel: chunks alternate between two buffer sets / streams, so that seeding and the estimator passes of one
    // chunk overlap the propagation of its neighbours
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_join = nullptr;
    // tardis_mc_progress (another host thread polls while propagate blocks): the running call's packet count, whether its kernel hands
    // packets out through next_packet over the whole call (wave kernel), whether it is complete; a stream and a pinned word of its own
    std::atomic<long long> progress_total{0};
    std::atomic<bool> progress_wave{false}, progress_done{true};
    std::mutex progress_mutex;
    hipStream_t stream_progress = nullptr;
    hipEvent_t ev_progress_reset = nullptr;  // recorded behind the reset of next_packet at the start of a call
    unsigned long long *progress_host = nullptr;
    // CU partition (option pass_cus: CUs per XCD set aside for the estimator passes, 0 = off): the propagation launches of a call of several
    // epochs run on a stream whose queue is masked to the other CUs, the passes of epoch k beside epoch k + 1 on a stream masked to these
    int pass_cus = 0, pass_cus_built = 0;
    hipStream_t stream_prop_m = nullptr, stream_pass_m = nullptr;
    hipEvent_t ev_fork_m = nullptr, ev_join_m = nullptr;
    DevBuf log_records[2], log_keys[2], log_cursor[2], log_bins[2], log_sorted[2], wave_cold_dev;
    DevBuf seed_chk[2], vp_scratch[2];  // (vp_scratch: per buffer set -- chunks on the two streams overlap)
    DevBuf vp_park;                     // pooled volleys with carry-over: one parked v-packet per lane
    int vp_carry_min_active = 16;       // pooled volleys: leave the volley phase once nothing waits and this few lanes still trace (0: never).  16: -7 % on the
                                        // configs[4] shape, -5 % on the tardis_example shape + 10 v-packets, flat from 16 to 32 (profiles/r05_volley_carry.txt)
    // wave kernel: word 397 of every packet's init_genrand sequence (lazy MT19937 seeding)
    double last_post_ms = 0.0;  // estimator passes (binning + accumulation) of the last propagate call
    double traces_per_packet = 0.0;  // measured by the last propagate (sizes the line-visit log of the next one)
    double log_budget_per_packet = 128.0;  // log records reserved per packet
    unsigned long long *events_host = nullptr;  // pinned: {events counter of the last propagate, its packet count}
    hipEvent_t ev_events = nullptr;
    std::vector<mc::WaveCold> wave_cold_host;
    // wave kernel: a propagate call is a sequence of epochs over one packet supply (LaneSave, propagate_wave.hpp)
    DevBuf lane_save, wave_save, suspended_dev;
    // drain compaction (option drain_compact = T: waves suspend once the supply has run out and T or fewer of their lanes are left; the live lanes are packed into
    // full waves and the rest of the call is a launch of fewer waves, beside the estimator passes): the packed grid's buffers, two sets for repeated packing
    int drain_compact = 0;
    int drain_pack_lanes = 64;  // live lanes per packed wave
    int est_one_level = 1;      // the passes' partition in one pass where the (shell, tile) bins are few enough to be ranked in LDS at once (0: always two levels)
    DevBuf lane_save_c[2], wave_save_c[2], seeded_states_c[2], drain_census;
    int compactions = 0;  // of the last propagate call
    DevBuf vq_req, vq_items, vq_count, vq_jsave;  // volley queue (variant 4, propagate_wave.hpp: VolleyRequest)
    long long vq_min_items = -1;  // switch the queue off for the rest of a call once a launch requests fewer v-packets (-1: automatic)
    int vq_min_active = 8, vq_oversubscribe = 4, vq_tracer_waves_per_simd = 6;
    unsigned *suspended_host = nullptr;  // pinned
    hipEvent_t ev_post[4] = {nullptr, nullptr, nullptr, nullptr};  // start / end of the estimator passes on log buffer set 0 / 1
    bool wave_epoch_mode = false, post_pending[2] = {false, false}, prop_pending = false;
    double sum_seed_ms = 0.0, sum_prop_ms = 0.0, sum_post_ms = 0.0;
    int launches = 0;
    int log_sets = 0;  // 2: the estimator passes of an epoch run beside the next launch (two log sets); 1: before it (one set); 0: 1 for calls of several epochs, 2 otherwise
    int last_variant = -1;  // kernel of the last propagate call (see tardis_mc_last_variant)
    int waves_per_simd = 4;  // register budget hint of the cooperative kernel (2: 256 VGPRs, 3: 168, 4: 128)
    // result streaming (tardis_mc_stream_results): the caller's per-packet arrays, what has been copied to them while the call ran, and the packets that were
    // in flight when their range was copied (their results are sent again by get_results)
    struct ResultStream {
        bool armed = false;        // destinations registered for the next propagate
        bool valid = false;        // the last propagate streamed: [0, upto) + the late list are (about to be) in the caller's arrays
        void *dst[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        long long upto = 0;        // packets [0, upto) were copied at launch boundaries
        long long n_late = 0;      // entries of the late list (may hold a packet more than once)
        unsigned late_capacity = 0;
    } rs;
    DevBuf rs_late, rs_late_count, rs_vals;
    hipStream_t rs_stream = nullptr;
    hipEvent_t rs_ev = nullptr;
    unsigned long long *rs_next_host = nullptr;  // pinned: the packet counter after a launch
    long long rs_min_packets = 1000000;  // a range worth sixteen copies of its own (option stream_min_packets; tests lower it)
    // RCCL
    void *comm = nullptr;
    int rank = 0, world = 1;
};

namespace {

int fail(TardisMcContext *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                      \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return fail(ctx, TARDIS_MC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                              \
    } while (0)

// [rows, cols] row-major -> [cols, rows] row-major (tile through LDS so both sides are coalesced)
__global__ void transpose_kernel(const double *__restrict__ in, double *__restrict__ out, long long rows, long long cols)
{
    __shared__ double tile[32][33];
    long long bx = (long long)blockIdx.x * 32, by = (long long)blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        long long r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        long long c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[c * rows + r] = tile[threadIdx.x][j];
    }
}

// nt[s * stride + l] = {nu_line[l], tau_t[s][l]} (the interleaved sweep table).  The frequency slot of the LAST line of the list, of the entries past a row's L
// lines and of the slack behind the last row holds -inf: the lane sweep's lean no-stop proof fails there (X = +inf) and hands the line to the exact evaluation,
// which never reads the last line's frequency (its distance is MISS_DISTANCE) -- so the sweep needs no per-line test for the end of the list.  `negative` is
// raised if an optical depth is negative or NaN: the lean proof assumes tau >= 0 (the host then keeps such tables on the separate-table kernels).
__global__ void __launch_bounds__(256) interleave_kernel(const double *__restrict__ nu_line, const double *__restrict__ tau_t, double2 *__restrict__ nt,
                                                         long long L, long long S, long long stride, long long total, int *negative)
{
    bool neg = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long s = i / stride, l = i - s * stride;
        const bool in = s < S && l < L;
        const double tau = in ? tau_t[s * L + l] : 0.0;
        neg |= !(tau >= 0.0);
        nt[i] = make_double2((in && l < L - 1) ? nu_line[l] : -__builtin_inf(), tau);
    }
    if (neg) atomicOr(negative, 1);
}

// Stores a launch-argument block into device memory.  The value travels in the kernel's argument buffer, which the runtime
// copies when the launch is enqueued -- unlike hipMemcpyAsync from pageable host memory, nothing on the host has to stay
// alive or unchanged afterwards, so back-to-back propagate calls need no stream synchronisation between them.
template <typename T>
__global__ void store_value_kernel(T *dst, const T v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
template <typename T>
hipError_t store_value(hipStream_t st, T *dst, const T &v)
{
    static_assert(sizeof(T) <= 3584, "argument block too large for the kernel-argument buffer");
    hipLaunchKernelGGL(store_value_kernel<T>, dim3(1), dim3(64), 0, st, dst, v);
    return hipGetLastError();
}
struct FirstErrorInit { long long v[2]; };

// sum private copies into copy 0 (in place)
__global__ void reduce_copies_kernel(double *base, long long n, long long stride, int copies)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        double s = base[i];
        for (int c = 1; c < copies; ++c) { s += base[i + c * stride]; base[i + c * stride] = 0.0; }
        base[i] = s;
    }
}

// diagnostics: element-wise device arithmetic for the numerics parity tests
__global__ void debug_eval_kernel(int op, const double *x, const double *y, double *out, long long n, uint32_t *scratch)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 7) {  // MT19937 stream of seed (uint32)x[0]
        if (i == 0) {
            mc::Rng rng;
            rng.seed(scratch, (uint32_t)x[0]);
            for (long long k = 0; k < n; ++k) out[k] = rng.random();
        }
        return;
    }
    if (i >= n) return;
    double a = x[i], b = y ? y[i] : 0.0, r;
    switch (op) {
    case 0: r = a + b; break;
    case 1: r = a * b; break;
    case 2: r = a / b; break;
    case 3: r = sqrt(a); break;
    case 4: r = mcm::log(a); break;
    case 5: r = mcm::exp(a); break;
    case 6: r = a * b + a; break;  // must NOT be contracted into an fma
    case 8: r = floor(a); break;
    case 9: r = ((mc::mid_range(a) || a == 0.0) && mc::mid_range(b)) ? mc::exact_div<true>(a, b, 1.0 / b) : a / b; break;
    default: r = 0.0;
    }
    out[i] = r;
}

// ---- micro-benchmarks of the memory system (design input; not part of the product path)
// which: 0 random fp64 atomic add, agent scope; 1 same, workgroup scope inside a per-XCD private slice;
//        2 fp64 atomic add, 16 consecutive doubles per 16-lane group, agent scope; 3 same, workgroup scope/XCD slice;
//        4 random 8-byte loads; 5 16-lane-coalesced 8-byte loads;
//        6..9 every lane reads its own random, naturally aligned block of 16 / 32 / 64 / 128 bytes (dwordx4 loads);
//        10 dependent chain: the address of a lane's next random 8-byte load comes out of the loaded value (latency);
//        11 / 12 lane-private vs quad-shared 64-byte blocks (see below)
//        13 / 14 packet hand-over through binned queues, 128- / 64-byte records (see below)
__global__ void microbench_kernel(int which, double *table, long long n, int iters, double *sink)
{
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(gtid + 1);
    const int lane16 = threadIdx.x & 15;
    const bool xcd_slice = (which == 1 || which == 3);
    long long span = n, base0 = 0;
    if (xcd_slice) { span = n / 8; base0 = span * (mc::xcc_id() & 7); }
    double acc = 0.0;
    if (which >= 6 && which <= 9) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int quads = 1 << (which - 6);  // 16-byte pieces per block
        const unsigned long long n_blocks = (unsigned long long)n / (2ull * quads);
        for (int it = 0; it < iters; ++it) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const v2d *b = reinterpret_cast<const v2d *>(table) + ((st >> 20) % n_blocks) * quads;
            v2d v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < quads) v[q] = b[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < quads) acc += v[q].x + v[q].y;
        }
    } else if (which == 11 || which == 12) {
        // 11: every lane reads its own random 64-byte block with four 16-byte loads (the lane sweeps' pattern);
        // 12: the same blocks, but the four lanes of a quad read ONE block per instruction, lane j its j-th 16 bytes, four
        //     instructions for the quad's four blocks (same bytes, same lines; fewer distinct lines per instruction)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const unsigned long long n_blocks = (unsigned long long)n / 8ull;
        const int lane = threadIdx.x & 63, j = lane & 3;
        for (int it = 0; it < iters; ++it) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned long long mine = (st >> 20) % n_blocks;
            v2d v[4];
            if (which == 11) {
                const v2d *b = reinterpret_cast<const v2d *>(table) + mine * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = b[q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned lo = (unsigned)__shfl((int)(unsigned)mine, (lane & ~3) + q, 64), hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), (lane & ~3) + q, 64);
                    const unsigned long long blk = ((unsigned long long)hi << 32) | lo;
                    v[q] = (reinterpret_cast<const v2d *>(table) + blk * 4)[j];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y;
        }
    } else if (which == 13 || which == 14) {
        // packet hand-over through binned queues (what re-binning the lanes of a wave by (shell, line tile) would cost per event,
        // profiles/r04_locality.txt): a lane appends its packet -- a 128-byte (13) or 64-byte (14) record -- to the queue of a
        // random one of 4900 bins (one returning atomic on the bin's cursor, 16-byte stores), and takes over a packet from a
        // random slot of another bin (16-byte loads).  table = [4900 cursors | 4900 queues of `cap` records].
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int rec_v2 = which == 13 ? 8 : 4;  // 16-byte pieces per record
        const unsigned long long n_bins = 4900ull;
        const unsigned long long cap = ((unsigned long long)n - 8192ull) / (n_bins * 2ull * (unsigned long long)rec_v2);
        unsigned long long *cursor = reinterpret_cast<unsigned long long *>(table);
        v2d *queues = reinterpret_cast<v2d *>(table + 8192);
        for (int it = 0; it < iters; ++it) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned long long b_out = (st >> 20) % n_bins, b_in = (st >> 40) % n_bins;
            const unsigned long long slot = __hip_atomic_fetch_add(&cursor[b_out], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % cap;
            v2d *dst = queues + (b_out * cap + slot) * rec_v2;
            const v2d rec = {acc, (double)it};
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < rec_v2) dst[q] = rec;
            const v2d *src = queues + (b_in * cap + (st >> 7) % cap) * rec_v2;
            v2d v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < rec_v2) v[q] = src[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q < rec_v2) acc += v[q].x + v[q].y;
        }
    } else if (which == 10) {
        unsigned long long r = st >> 20;
        for (int it = 0; it < iters; ++it) {
            const double v = table[(long long)(r % (unsigned long long)span)];
            r = r * 6364136223846793005ull + 1442695040888963407ull + (unsigned long long)__double_as_longlong(v);
            acc += v;
        }
    } else
    for (int it = 0; it < iters; ++it) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        unsigned long long r = st >> 20;
        long long idx;
        if (which == 2 || which == 3 || which == 5) {
            unsigned long long rg = __shfl(r, threadIdx.x & ~15, 64);  // group-uniform random base
            idx = base0 + (long long)((rg % (unsigned long long)(span / 16)) * 16) + lane16;
        } else
            idx = base0 + (long long)(r % (unsigned long long)span);
        if (which == 0 || which == 2) __hip_atomic_fetch_add(&table[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (which == 1 || which == 3) __hip_atomic_fetch_add(&table[idx], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else acc += table[idx];
    }
    if (acc == 123.456) sink[0] = acc;
}

// which = 15: the box's streaming rate -- a wide coalesced copy of the table's first half onto its second (16 bytes per lane and access, grid-stride):
// n_doubles x 8 bytes cross the memory interface per pass (half read, half written); bench.py reports it as roofline.peak_measured
__global__ void __launch_bounds__(256) stream_copy_kernel(double *table, long long n, int iters)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const long long half = (n / 4) * 2;  // doubles per half, whole 16-byte pieces
    const v2d *__restrict__ src = reinterpret_cast<const v2d *>(table);
    v2d *__restrict__ dst = reinterpret_cast<v2d *>(table + half);
    const long long pieces = half / 2, step = (long long)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * step < pieces; i += 4 * step) {  // four 16-byte pieces in flight per lane
            const v2d a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + step);
            const v2d c = __builtin_nontemporal_load(src + i + 2 * step), d = __builtin_nontemporal_load(src + i + 3 * step);
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + step);
            __builtin_nontemporal_store(c, dst + i + 2 * step); __builtin_nontemporal_store(d, dst + i + 3 * step);
        }
        for (; i < pieces; i += step) dst[i] = src[i];
    }
}

// ---- real-packet spectrum and filtered luminosities from the resident per-packet outputs
// (SpectrumSolver.montecarlo_emitted/reabsorbed_luminosity, tardis/spectrum/base.py:140-159: np.histogram with the
//  spectrum_frequency_grid edges, weights = +/- output_energies / time_of_simulation split on output_energies >= 0
//  (montecarlo_transport_state.py:130-160); calculate_filtered_luminosity, tardis/spectrum/luminosity.py:5-30)
__global__ void spectrum_kernel(const double *__restrict__ out_nu, const double *__restrict__ out_e, long long n,
                                const double *__restrict__ edges, int n_edges, double t_sim, double nu_start, double nu_end,
                                double *hist_emitted, double *hist_reabsorbed, double *lum /* [2] */)
{
    const int B = n_edges - 1;
    const double e0 = edges[0], eN = edges[B];
    const double inv_delta = (double)B / (eN - e0);
    double lum_e = 0.0, lum_r = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double e = out_e[i], nu = out_nu[i];
        const bool emitted = e >= 0;
        const double l = emitted ? (e / t_sim) : -(e / t_sim);
        if (nu > nu_start && nu < nu_end) { if (emitted) lum_e += l; else lum_r += l; }
        if (nu >= e0 && nu <= eN) {
            // numpy: bin k holds edges[k] <= x < edges[k+1], the last bin is closed on the right; start from the uniform
            // estimate and pin it with the actual edge values
            int k = (int)((nu - e0) * inv_delta);
            k = k < 0 ? 0 : (k > B - 1 ? B - 1 : k);
            while (k > 0 && nu < edges[k]) --k;
            while (k < B - 1 && nu >= edges[k + 1]) ++k;
            mc::atomic_add_f64(emitted ? &hist_emitted[k] : &hist_reabsorbed[k], l);
        }
    }
    // block reduction of the two luminosity sums
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = lum_e; sh[1][threadIdx.x] = lum_r;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { mc::atomic_add_f64(&lum[0], sh[0][0]); mc::atomic_add_f64(&lum[1], sh[1][0]); }
}

// ---- radiation-field update from the resident estimators (MCRadiationFieldPropertiesSolver.solve,
// tardis/transport/montecarlo/estimators/mc_rad_field_solver.py:37-144; intensity_black_body, tardis/util/base.py:279-302)
struct RadFieldConsts { double t_rad_const, four_sigma, jblue_norm_num, four_pi_tsim, tsim, planck_coef, h, k_b, w_epsilon, c_ang; };

__global__ void radfield_shell_kernel(const double *J, const double *nubar, const double *volume, int S, RadFieldConsts k,
                                      double *t_rad, double *w, double *norm)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double t = k.t_rad_const * nubar[s] / J[s];
    t_rad[s] = t;
    const double t2 = t * t;
    w[s] = J[s] / (k.four_sigma * (t2 * t2) * k.tsim * volume[s]);
    norm[s] = k.jblue_norm_num / (k.four_pi_tsim * volume[s]);
}

__global__ void radfield_jblue_kernel(const double *__restrict__ jblue_t, const double *__restrict__ nu_line,
                                      const double *__restrict__ t_rad, const double *__restrict__ w,
                                      const double *__restrict__ norm, int S, long long L, RadFieldConsts k, int optical_window,
                                      double *__restrict__ out_t)
{
    const int s = blockIdx.y;
    const double beta = 1 / (k.k_b * t_rad[s]), ws = w[s], ns = norm[s];
    for (long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x; l < L; l += (long long)gridDim.x * blockDim.x) {
        const double nu = nu_line[l];
        const double est = jblue_t[(long long)s * L + l] * ns;
        double value = est;
        bool outside = false;
        if (optical_window) {
            const double wav = k.c_ang / nu;  // Angstrom
            outside = !(wav > 2500.0 && wav < 10000.0);
        }
        if (est == 0.0 || outside) {
            const double planck = ws * (k.planck_coef * (nu * nu * nu) / (mcm::exp(k.h * nu * beta) - 1));
            value = outside ? planck : value;
            if (est == 0.0) value = k.w_epsilon * planck;
        }
        out_t[(long long)s * L + l] = value;
    }
}

hipError_t launch_transpose(hipStream_t s, const double *in, double *out, long long rows, long long cols)
{
    dim3 block(32, 8), grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, s, in, out, rows, cols);
    return hipGetLastError();
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EstLayout { size_t J, nubar, vhist, jblue, edot, copy_stride, reduce_elems, total; };
EstLayout est_layout(size_t S, size_t L, size_t G, int copies)
{
    EstLayout e;
    e.J = 0;
    e.nubar = S;
    e.vhist = 2 * S;
    size_t head = align_up(2 * S + G, 32);
    e.jblue = head;
    e.edot = head + S * L;
    e.copy_stride = 2 * S * L;             // distance between copy c and c+1 of the same table
    e.reduce_elems = head + 2 * S * L;     // contiguous range that is all-reduced (copy 0)
    e.total = head + (size_t)copies * 2 * S * L;
    return e;
}

int ensure_estimators(TardisMcContext *ctx)
{
    if (!ctx->have_opacity || !ctx->have_config) return TARDIS_MC_OK;
    size_t S = ctx->n_shells, L = ctx->n_lines, G = (size_t)ctx->cfg.n_spectrum_grid;
    if (ctx->est_valid && ctx->est_S == S && ctx->est_L == L && ctx->est_G == G) return TARDIS_MC_OK;
    EstLayout e = est_layout(S, L, G, ctx->est_copies);
    HIP_TRY(ctx, ctx->est.ensure(e.total * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->est.p, 0, e.total * sizeof(double), ctx->stream));
    ctx->est_S = S; ctx->est_L = L; ctx->est_G = G;
    ctx->est_valid = true;
    return TARDIS_MC_OK;
}

template <typename T>
int upload(TardisMcContext *ctx, DevBuf &buf, const T *host, size_t n)
{
    HIP_TRY(ctx, buf.ensure(n * sizeof(T)));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return TARDIS_MC_OK;
}

// ---- boundary copies between the caller's (pageable) arrays and HBM: plain hipMemcpy.  (Round 6 measured a pipeline of its own -- four worker threads, each with a
// stream and two pinned 8-MB buffers -- against it: no gain, 29.7 vs 31.6 ms for the 1.28 GB of per-packet results of a 1e7-packet call, 10-26 vs 9.6 ms for the
// packet upload; the runtime's own staging already runs at ~40-50 GB/s once the destination pages exist.  What did cost 160 ms of that call was on the Python side:
// fresh tracker arrays allocated, page-faulted and copied once more into the caller's -- Engine.get_results now writes into the caller's arrays;
// profiles/r06_boundary.txt.)
struct CopyJob { void *host; void *dev; size_t bytes; };

hipError_t host_copy(TardisMcContext *ctx, const std::vector<CopyJob> &jobs, bool to_device)
{
    (void)ctx;
    for (const CopyJob &j : jobs) {
        if (!j.host || !j.dev || !j.bytes) continue;
        hipError_t e = to_device ? hipMemcpy(j.dev, j.host, j.bytes, hipMemcpyHostToDevice) : hipMemcpy(j.host, j.dev, j.bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// the sixteen per-packet result arrays on the device / in a TardisMcResult, in one order: out_nu, out_e, nine tracker doubles, five tracker integers
void per_packet_device_arrays(TardisMcContext *ctx, void *dev[16])
{
    dev[0] = ctx->out_nu.p; dev[1] = ctx->out_e.p;
    for (int k = 0; k < 9; ++k) dev[2 + k] = ctx->track ? ctx->li_f64[k].p : nullptr;
    for (int k = 0; k < 5; ++k) dev[11 + k] = ctx->track ? ctx->li_i64[k].p : nullptr;
}
void per_packet_host_arrays(const TardisMcResult *r, void *host[16])
{
    host[0] = r->output_nus; host[1] = r->output_energies;
    double *f[] = {r->li_radius, r->li_nu, r->li_energy, r->li_before_nu, r->li_before_mu, r->li_before_energy, r->li_after_nu, r->li_after_mu, r->li_after_energy};
    for (int k = 0; k < 9; ++k) host[2 + k] = f[k];
    int64_t *g[] = {r->li_shell_id, r->li_interaction_type, r->li_line_absorb_id, r->li_line_emit_id, r->li_interactions_count};
    for (int k = 0; k < 5; ++k) host[11 + k] = g[k];
}

__global__ void __launch_bounds__(256) narrow_seeds_kernel(const long long *__restrict__ in, long long n, uint32_t *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

mc::DeviceProblem make_device_problem(TardisMcContext *ctx)
{
    mc::DeviceProblem P{};
    P.n_packets = ctx->n_packets;
    P.r0 = ctx->r0.as<double>(); P.mu0 = ctx->mu0.as<double>(); P.nu0 = ctx->nu0.as<double>(); P.e0 = ctx->e0.as<double>();
    P.seeds = ctx->seeds.as<uint32_t>();
    P.out_nu = ctx->out_nu.as<double>(); P.out_e = ctx->out_e.as<double>();
    if (ctx->track) {
        double **f[] = {&P.li_radius, &P.li_nu, &P.li_energy, &P.li_before_nu, &P.li_before_mu, &P.li_before_energy,
                        &P.li_after_nu, &P.li_after_mu, &P.li_after_energy};
        for (int k = 0; k < 9; ++k) *f[k] = ctx->li_f64[k].as<double>();
        long long **g[] = {&P.li_shell_id, &P.li_interaction_type, &P.li_line_absorb_id, &P.li_line_emit_id,
                           &P.li_interactions_count};
        for (int k = 0; k < 5; ++k) *g[k] = ctx->li_i64[k].as<long long>();
        P.li_rec = ctx->li_rec.as<uint4>();
    }
    P.n_shells = ctx->n_shells;
    P.r_inner = ctx->r_inner.as<double>(); P.r_outer = ctx->r_outer.as<double>();
    P.t_exp = ctx->t_exp;
    P.n_lines = ctx->n_lines; P.n_trans = ctx->n_trans;
    P.nu_line = ctx->nu_line.as<double>(); P.tau_t = ctx->tau_t.as<double>(); P.n_e = ctx->n_e.as<double>();
    P.prob_t = ctx->prob_t.as<double>();
    P.line2level = ctx->line2level.as<int>(); P.block_edge = ctx->block_edge.as<int>(); P.ttype = ctx->ttype.as<int>();
    P.dest = ctx->dest.as<int>(); P.tline = ctx->tline.as<int>();
    EstLayout e = est_layout(ctx->est_S, ctx->est_L, ctx->est_G, ctx->est_copies);
    double *base = ctx->est.as<double>();
    P.J = base + e.J; P.nubar = base + e.nubar; P.vhist = base + e.vhist;
    P.jblue_t = base + e.jblue; P.edot_t = base + e.edot;
    P.est_copy_stride = (long long)e.copy_stride;
    P.n_est_copies = ctx->est_copies;
    const TardisMcConfig &c = ctx->cfg;
    P.line_interaction_type = c.line_interaction_type;
    P.disable_line_scattering = c.disable_line_scattering;
    P.n_vpackets = c.number_of_vpackets;
    P.survival_probability = c.survival_probability;
    P.tau_russian = c.vpacket_tau_russian;
    P.spawn_start = c.vpacket_spawn_start_frequency;
    P.spawn_end = c.vpacket_spawn_end_frequency;
    P.sigma_thomson = c.sigma_thomson;
    P.grid = ctx->grid.as<double>();
    P.n_grid = (int)c.n_spectrum_grid;
    if (c.n_spectrum_grid >= 2) {
        P.grid0 = ctx->grid_host[0];
        P.grid_last = ctx->grid_host[c.n_spectrum_grid - 1];
        P.delta_nu = ctx->grid_host[1] - ctx->grid_host[0];
    }
    if (c.enable_vpacket_tracking && c.number_of_vpackets > 0 && ctx->vlog_capacity > 0) {
        P.vlog_count = ctx->vlog_count.as<unsigned long long>();
        P.vlog_capacity = ctx->vlog_capacity;
        P.vlog_packet = ctx->vlog_packet.as<long long>(); P.vlog_seq = ctx->vlog_seq.as<int>();
        P.vlog_nu = ctx->vlog_nu.as<double>(); P.vlog_energy = ctx->vlog_energy.as<double>();
        P.vlog_mu = ctx->vlog_mu.as<double>(); P.vlog_r = ctx->vlog_r.as<double>();
    }
    P.rng_state = ctx->rng_state.as<uint32_t>();
    P.counters = ctx->counters.as<unsigned long long>();
    P.first_error = ctx->first_error.as<long long>();
    P.next_packet = ctx->next_packet.as<unsigned long long>();
    P.debug_flags = ctx->debug_flags;
    return P;
}

template <bool FULL, bool VPK>
void launch_lane(TardisMcContext *ctx, const mc::DeviceProblem &P, int blocks, size_t lds)
{
    if (ctx->track)
        hipLaunchKernelGGL((mc::propagate_lane_kernel<FULL, VPK, true>), dim3(blocks), dim3(256), lds, ctx->stream, P);
    else
        hipLaunchKernelGGL((mc::propagate_lane_kernel<FULL, VPK, false>), dim3(blocks), dim3(256), lds, ctx->stream, P);
}

}  // namespace

extern "C" {

int tardis_mc_abi_version(void) { return TARDIS_MC_ABI_VERSION; }

int tardis_mc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tardis_mc_create(int device_id, TardisMcContext **out_ctx)
{
    if (!out_ctx) return fail(nullptr, TARDIS_MC_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, TARDIS_MC_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, TARDIS_MC_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, n);
    TardisMcContext *ctx = new TardisMcContext();
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess || (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) {
        int rc = fail(nullptr, TARDIS_MC_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    if (const char *v = getenv("TARDIS_MC_VARIANT")) ctx->variant = atoi(v);
    if (const char *v = getenv("TARDIS_MC_BLOCKS_PER_CU")) ctx->blocks_per_cu = std::max(1, atoi(v));
    if (const char *v = getenv("TARDIS_MC_DEBUG_FLAGS")) ctx->debug_flags = atoi(v);
    if (const char *v = getenv("TARDIS_MC_EST_PIPELINE")) ctx->est_pipeline = atoi(v) ? 1 : 0;
    if (const char *v = getenv("TARDIS_MC_EST_COPIES")) ctx->est_copies = std::max(1, std::min(8, atoi(v)));
    *out_ctx = ctx;
    return TARDIS_MC_OK;
}

void tardis_mc_destroy(TardisMcContext *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    DevBuf *all[] = {&ctx->r_inner, &ctx->r_outer, &ctx->nu_line, &ctx->tau_t, &ctx->n_e, &ctx->prob_t, &ctx->cum_t, &ctx->trans_nu, &ctx->line2level,
                     &ctx->block_edge, &ctx->ttype, &ctx->dest, &ctx->tline, &ctx->staging, &ctx->line_block, &ctx->trans_rec, &ctx->bucket_first, &ctx->est, &ctx->grid, &ctx->r0,
                     &ctx->mu0, &ctx->nu0, &ctx->e0, &ctx->seeds, &ctx->out_nu, &ctx->out_e, &ctx->vlog_count,
                     &ctx->vlog_packet, &ctx->vlog_seq, &ctx->vlog_nu, &ctx->vlog_energy, &ctx->vlog_mu, &ctx->vlog_r,
                     &ctx->rng_state, &ctx->counters, &ctx->first_error, &ctx->next_packet, &ctx->seeded_states, &ctx->problem_dev};
    for (DevBuf *b : all) b->release();
    for (int b = 0; b < 2; ++b) {
        ctx->log_records[b].release(); ctx->log_keys[b].release(); ctx->log_cursor[b].release(); ctx->log_bins[b].release();
        ctx->log_sorted[b].release();
    }
    ctx->log_part.release();
    ctx->wave_cold_dev.release();
    ctx->seed_chk[0].release(); ctx->seed_chk[1].release(); ctx->vp_scratch[0].release(); ctx->vp_scratch[1].release(); ctx->vp_park.release();
    for (auto &b : ctx->li_f64) b.release();
    for (auto &b : ctx->li_i64) b.release();
    ctx->li_rec.release();
    ctx->cum16.release(); ctx->rec16.release(); ctx->quad_info.release(); ctx->line_block_c.release();
    ctx->hot_sec.release(); ctx->hot_mass.release(); ctx->hot_flag.release(); ctx->blk_tab.release();
    ctx->tau_pfx.release(); ctx->tau_rowsum.release(); ctx->pfx_flag.release(); ctx->nt_t.release();
    ctx->lane_save.release(); ctx->wave_save.release(); ctx->suspended_dev.release();
    for (int k = 0; k < 2; ++k) { ctx->lane_save_c[k].release(); ctx->wave_save_c[k].release(); ctx->seeded_states_c[k].release(); }
    ctx->drain_census.release();
    ctx->vq_req.release(); ctx->vq_items.release(); ctx->vq_count.release(); ctx->vq_jsave.release();
    if (ctx->suspended_host) (void)hipHostFree(ctx->suspended_host);
    for (hipEvent_t e : ctx->ev_post) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_chunk) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_tune) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_split) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->events_host) (void)hipHostFree(ctx->events_host);
    if (ctx->ev_events) (void)hipEventDestroy(ctx->ev_events);
    ctx->rs_late.release(); ctx->rs_late_count.release(); ctx->rs_vals.release();
    if (ctx->rs_stream) (void)hipStreamDestroy(ctx->rs_stream);
    if (ctx->rs_ev) (void)hipEventDestroy(ctx->rs_ev);
    if (ctx->rs_next_host) (void)hipHostFree(ctx->rs_next_host);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_progress) (void)hipStreamDestroy(ctx->stream_progress);
    if (ctx->ev_progress_reset) (void)hipEventDestroy(ctx->ev_progress_reset);
    if (ctx->progress_host) (void)hipHostFree(ctx->progress_host);
    if (ctx->stream_prop_m) (void)hipStreamDestroy(ctx->stream_prop_m);
    if (ctx->stream_pass_m) (void)hipStreamDestroy(ctx->stream_pass_m);
    if (ctx->ev_fork_m) (void)hipEventDestroy(ctx->ev_fork_m);
    if (ctx->ev_join_m) (void)hipEventDestroy(ctx->ev_join_m);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *tardis_mc_last_error(const TardisMcContext *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tardis_mc_set_option(TardisMcContext *ctx, const char *name, long long value)
{
    if (!ctx || !name) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    std::string n(name);
    if (n == "variant") ctx->variant = (int)value;
    else if (n == "blocks_per_cu") ctx->blocks_per_cu = std::max(1, (int)value);
    else if (n == "track_last_interaction") ctx->track = value != 0;
    else if (n == "estimator_copies") { ctx->est_copies = std::max(1, std::min(8, (int)value)); ctx->est_valid = false; }
    else if (n == "vpacket_log_capacity") { ctx->vlog_capacity = value; ctx->vlog_capacity_user = value > 0; }
    else if (n == "debug_flags") ctx->debug_flags = (int)value;
    else if (n == "drain_split") ctx->drain_split = value ? 1 : 0;
    else if (n == "drain_compact") ctx->drain_compact = (int)std::max<long long>(0, std::min<long long>(48, value));
    else if (n == "est_one_level") ctx->est_one_level = value ? 1 : 0;
    else if (n == "drain_pack_lanes") ctx->drain_pack_lanes = (int)std::max<long long>(1, std::min<long long>(64, value));
    else if (n == "walk_sector_packing") ctx->walk_sector_packing = value ? 1 : 0;
    else if (n == "walk_hot") ctx->walk_hot = value < 0 ? -1 : (value ? 1 : 0);  // (like walk_sector_packing: before set_opacity)
    else if (n == "walk_hot_min_mass") ctx->walk_hot_min_mass = (int)std::max<long long>(0, std::min<long long>(value, 1001));
    else if (n == "walk_hot_min_mass_long") ctx->walk_hot_min_mass_long = (int)std::max<long long>(0, std::min<long long>(value, 1001));
    else if (n == "vpacket_screening") ctx->vpacket_screening = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "waves_per_simd") ctx->waves_per_simd = (int)value;
    else if (n == "lane_sweep_min_active") {  // (< 0: the automatic choice again)
        ctx->ls_min_active_user = value >= 0;
        ctx->ls_min_active = value >= 0 ? (int)std::min<long long>(value, 63) : 8;
    }
    else if (n == "lane_sweep_max_steps") ctx->ls_max_steps = (int)std::max<long long>(1, value);
    else if (n == "vq_oversubscribe") ctx->vq_oversubscribe = (int)std::max<long long>(1, std::min<long long>(value, 64));
    else if (n == "vq_tracer_waves_per_simd") ctx->vq_tracer_waves_per_simd = (int)std::max<long long>(1, std::min<long long>(value, 16));
    else if (n == "vq_min_items") ctx->vq_min_items = value;
    else if (n == "vq_min_active") ctx->vq_min_active = (int)std::max<long long>(0, std::min<long long>(value, 63));
    else if (n == "log_tail_split") ctx->log_tail_split = value ? 1 : 0;
    else if (n == "est_pipeline") ctx->est_pipeline = value ? 1 : 0;
    else if (n == "pass_cus") ctx->pass_cus = (int)std::max<long long>(0, std::min<long long>(value, 16));
    else if (n == "vpk_wave_min_packets") ctx->vpk_wave_min_packets = std::max<long long>(0, value);
    else if (n == "ls_waves_per_simd") { ctx->ls_waves_per_simd = (value == 3 || value == 4) ? (int)value : 0; ctx->ls_tune.n = -1; }
    else if (n == "epoch_split") ctx->epoch_split = value ? 1 : 0;
    else if (n == "log_by_shell") ctx->log_by_shell = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "stream_min_packets") ctx->rs_min_packets = std::max<long long>(64, value);
    else if (n == "sweep_table") ctx->sweep_table = (int)std::max<long long>(-1, std::min<long long>(value, 2));
    else if (n == "vpk_wide_registers") ctx->vpk_wide_registers = (int)std::max<long long>(0, std::min<long long>(value, 2));
    else if (n == "bucket_lines_permille") ctx->bucket_lines_permille = (int)std::max<long long>(50, std::min<long long>(value, 16000));
    else if (n == "vp_carry_min_active") ctx->vp_carry_min_active = (int)std::max<long long>(0, std::min<long long>(value, 63));
    else if (n == "est_accumulate") ctx->est_accumulate = (int)std::max<long long>(0, std::min<long long>(value, 3));
    else if (n == "log_tail_packets") ctx->log_tail_packets = (int)std::max<long long>(0, std::min<long long>(value, 1000));
    else if (n == "log_chunk_records") ctx->log_chunk_records = value <= 0 ? 0 : std::max<long long>(256, std::min<long long>(value, 1 << 20));
    else if (n == "walk_min_active") {  // (-1: never carry a walk over; < -1: the automatic choice again)
        ctx->walk_min_active_user = value >= -1;
        ctx->walk_min_active = value >= -1 ? (int)std::min<long long>(value, 63) : 8;
    }
    else if (n == "group_size") ctx->group_size = (value == 4 || value == 8 || value == 16) ? (int)value : 0;
    else if (n == "pipeline_chunks") {}  // (round 1: chunks on two streams; a call of the wave kernel now runs as epochs -- accepted, ignored)
    else if (n == "log_capacity") { ctx->log_capacity = std::max<long long>(0, value); ctx->log_capacity_user = true; }
    else if (n == "log_sets") ctx->log_sets = (value == 1 || value == 2) ? (int)value : 0;  // 1: the estimator passes of an epoch run before the next epoch, not beside it; 0: automatic
    else if (n == "chunk_packets") ctx->chunk_packets = std::max<long long>(1024, value);
    else return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
    return TARDIS_MC_OK;
}

int tardis_mc_set_geometry(TardisMcContext *ctx, const TardisMcGeometry *g)
{
    if (!ctx || !g || g->n_shells <= 0 || !g->r_inner || !g->r_outer)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid geometry");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->r_inner, g->r_inner, (size_t)g->n_shells))) return rc;
    if ((rc = upload(ctx, ctx->r_outer, g->r_outer, (size_t)g->n_shells))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_shells = (int)g->n_shells;
    ctx->t_exp = g->time_explosion;
    ctx->have_geometry = true;
    return TARDIS_MC_OK;
}

int tardis_mc_set_opacity(TardisMcContext *ctx, const TardisMcOpacity *o)
{
    if (!ctx || !o || o->n_lines <= 0 || o->n_shells <= 0 || o->n_transitions <= 0 || !o->electron_density ||
        !o->line_list_nu || !o->tau_sobolev || !o->transition_probabilities)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid opacity state");
    if (o->n_lines > 0x7ffffff0LL || o->n_transitions > 0x7ffffff0LL)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "line / transition count exceeds 32-bit device indices");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool tmark_on = getenv("TARDIS_MC_TIME_OPACITY") != nullptr;  // (diagnostic: wall time of the stages of this call on stderr)
    auto tmark_t0 = std::chrono::steady_clock::now();
    auto tmark = [&](const char *what) {
        if (!tmark_on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "set_opacity: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tmark_t0).count());
        tmark_t0 = now;
    };
    const size_t L = (size_t)o->n_lines, S = (size_t)o->n_shells, T = (size_t)o->n_transitions;
    const size_t E = (size_t)o->n_macro_block_edges;
    // validate the macro-atom index tables on the host: the device walks them without bounds checks
    const bool macro = T > 1 || E > 1;
    if (macro) {
        if (!o->line2macro_level_upper || !o->macro_block_edge_index || !o->transition_type || !o->destination_level_id ||
            !o->transition_line_id || E < 2)
            return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "macro-atom tables missing");
        const long long n_levels = (long long)E - 1;
        for (size_t i = 0; i + 1 < E; ++i)
            if (o->macro_block_edge_index[i] < 0 || o->macro_block_edge_index[i] > o->macro_block_edge_index[i + 1] ||
                o->macro_block_edge_index[i + 1] > (long long)T)
                return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "macro_block_edge_index not monotone within [0,T]");
        for (size_t i = 0; i < L; ++i)
            if (o->line2macro_level_upper[i] < 0 || o->line2macro_level_upper[i] >= n_levels)
                return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "line2macro_level_upper[%zu] out of range", i);
        for (size_t i = 0; i < T; ++i) {
            if (o->transition_type[i] >= 0 && (o->destination_level_id[i] < 0 || o->destination_level_id[i] >= n_levels))
                return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "destination_level_id[%zu] out of range", i);
            if (o->transition_type[i] == -1 && (o->transition_line_id[i] < 0 || o->transition_line_id[i] >= (long long)L))
                return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "transition_line_id[%zu] out of range", i);
        }
    }
    tmark("validate");
    int rc;
    // (slack at the end of the line list and of the tau table: the lane sweep loads whole chunks, propagate_wave.hpp)
    HIP_TRY(ctx, ctx->nu_line.ensure((L + 2 * mc::LS_CHUNK) * sizeof(double)));
    if ((rc = upload(ctx, ctx->nu_line, o->line_list_nu, L))) return rc;
    if ((rc = upload(ctx, ctx->n_e, o->electron_density, S))) return rc;
    // tau [L,S] -> [S][L]
    HIP_TRY(ctx, ctx->staging.ensure(std::max(L, T) * S * sizeof(double)));
    HIP_TRY(ctx, ctx->tau_t.ensure((L * S + 2 * mc::LS_CHUNK) * sizeof(double)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, host_copy(ctx, {{(void *)o->tau_sobolev, ctx->staging.p, L * S * sizeof(double)}}, true));
    HIP_TRY(ctx, launch_transpose(ctx->stream, ctx->staging.as<double>(), ctx->tau_t.as<double>(), (long long)L, (long long)S));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    tmark("nu, n_e, tau up + transpose");
    // probabilities [T,S] -> [S][T]
    HIP_TRY(ctx, ctx->prob_t.ensure(T * S * sizeof(double)));
    HIP_TRY(ctx, host_copy(ctx, {{(void *)o->transition_probabilities, ctx->staging.p, T * S * sizeof(double)}}, true));
    HIP_TRY(ctx, launch_transpose(ctx->stream, ctx->staging.as<double>(), ctx->prob_t.as<double>(), (long long)T, (long long)S));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    tmark("probabilities up + transpose");
    static const int64_t zero64 = 0;
    std::vector<int> idx32[5];  // (one staging vector per table: the copies are asynchronous, ONE synchronisation below covers them all)
    auto up32 = [&](int k, DevBuf &buf, const int64_t *host, size_t n) -> int {
        idx32[k].resize(n);
        for (size_t i = 0; i < n; ++i) idx32[k][i] = (int)host[i];
        return upload(ctx, buf, idx32[k].data(), n);
    };
    if ((rc = up32(0, ctx->line2level, macro ? o->line2macro_level_upper : &zero64, macro ? L : 1))) return rc;
    if ((rc = up32(1, ctx->block_edge, macro ? o->macro_block_edge_index : &zero64, macro ? E : 1))) return rc;
    if ((rc = up32(2, ctx->ttype, macro ? o->transition_type : &zero64, macro ? T : 1))) return rc;
    if ((rc = up32(3, ctx->dest, macro ? o->destination_level_id : &zero64, macro ? T : 1))) return rc;
    if ((rc = up32(4, ctx->tline, macro ? o->transition_line_id : &zero64, macro ? T : 1))) return rc;
    tmark("index tables int32 up");
    {   // packed macro-atom tables of the cooperative kernel
        std::vector<int> lb(2 * (macro ? L : 1), 0), rec(4 * (macro ? T : 1), 0);
        if (macro) {
            for (size_t i = 0; i < L; ++i) {
                const int64_t lvl = o->line2macro_level_upper[i];
                lb[2 * i] = (int)o->macro_block_edge_index[lvl];
                lb[2 * i + 1] = (int)o->macro_block_edge_index[lvl + 1];
            }
            for (size_t t = 0; t < T; ++t) {
                rec[4 * t] = (int)o->transition_line_id[t];
                rec[4 * t + 1] = (int)o->transition_type[t];
                if (o->transition_type[t] >= 0) {
                    const int64_t lvl = o->destination_level_id[t];
                    rec[4 * t + 2] = (int)o->macro_block_edge_index[lvl];
                    rec[4 * t + 3] = (int)o->macro_block_edge_index[lvl + 1];
                }
            }
        }
        if ((rc = upload(ctx, ctx->line_block, lb.data(), lb.size()))) return rc;
        if ((rc = upload(ctx, ctx->trans_rec, rec.data(), rec.size()))) return rc;
        // frequency of the line every emission transition ends in, next to its record (one round trip instead of two)
        std::vector<double> tnu(macro ? T : 1, 0.0);
        if (macro)
            for (size_t t = 0; t < T; ++t)
                if (o->transition_type[t] == -1) tnu[t] = o->line_list_nu[o->transition_line_id[t]];
        if ((rc = upload(ctx, ctx->trans_nu, tnu.data(), tnu.size()))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    tmark("packed tables lb / rec / tnu");
    {   // running sums of the transition probabilities within their blocks (macro_cumulative_kernel)
        HIP_TRY(ctx, ctx->cum_t.ensure((T * S + 8) * sizeof(double)));  // (+8: the jump search reads eight entries at a time)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->cum_t.p, ctx->prob_t.p, T * S * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->prob_negative = false;
        if (macro && E > 1) {
            HIP_TRY(ctx, ctx->pfx_flag.ensure(sizeof(int)));  // (a flag word the context keeps: no hipMalloc / hipFree per opacity state)
            int *flag = ctx->pfx_flag.as<int>();
            HIP_TRY(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
            const long long n = (long long)(E - 1) * (long long)S;
            hipLaunchKernelGGL(mc::macro_cumulative_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->prob_t.as<double>(),
                               ctx->cum_t.as<double>(), ctx->block_edge.as<int>(), (int)(E - 1), (long long)T, (int)S, flag);
            int neg = 0;
            hipError_t e1 = hipGetLastError();
            hipError_t e2 = hipMemcpyAsync(&neg, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
            hipError_t e3 = hipStreamSynchronize(ctx->stream);
            HIP_TRY(ctx, e1); HIP_TRY(ctx, e2); HIP_TRY(ctx, e3);
            ctx->prob_negative = neg != 0;
        }
    }
    tmark("running sums");
    ctx->have_walk_tables = false;
    ctx->have_hot = false;
    ctx->n_hot_blocks = 0;
    ctx->pfx_valid = false;  // (the prefix sums of the new tau table are built by the first propagate call that traces v-packets)
    ctx->nt_valid = false;   // (likewise the interleaved sweep table: by the first call that sweeps on it)
    if (macro && E > 1 && !ctx->prob_negative) {
        // compact tables of the per-lane macro-atom walk (walk_tables.hpp): blocks at 16-byte aligned compact offsets.  The walk is
        // bound by the number of memory requests, and a block's window of running sums is fetched in 64-byte sectors: a block of
        // up to 32 entries (one window) that would straddle a sector boundary starts at the next boundary instead (the skipped
        // quads are padding no block owns), longer blocks start on a boundary -- one request per jump instead of 1.5.
        const size_t n_levels = E - 1;
        std::vector<long long> c0(n_levels + 1);
        long long tc = 0;
        for (size_t b = 0; b < n_levels; ++b) {
            const long long len = (o->macro_block_edge_index[b + 1] - o->macro_block_edge_index[b] + 7) / 8 * 8;
            if (ctx->walk_sector_packing && len > 0) {
                const long long in_sector = tc & 31;  // (32 entries of 2 bytes per 64-byte sector)
                if (in_sector != 0 && (len > 32 || in_sector + len > 32)) tc += 32 - in_sector;
            }
            c0[b] = tc;
            tc += len;
        }
        c0[n_levels] = tc;
        const long long n_quads = tc / 8;
        const unsigned long long stride = ((unsigned long long)tc + 31ull) / 32ull * 32ull + mc::WALK_SLACK;  // (rows start on sector boundaries)
        if (tc > 0 && stride * S < (1ull << 32) && tc < (1LL << 30)) {
            // hot sectors (walk_tables.hpp): measured on the device (total width of a block's six widest intervals, per shell),
            // chosen here (mean over the shells), then built once more with the destinations' flags in place
            std::vector<unsigned char> hot(n_levels, 0);
            if (ctx->walk_hot != 0 && n_levels < (size_t)mc::WALK_HOT) {
                const long long nb = (long long)n_levels * (long long)S;
                HIP_TRY(ctx, ctx->hot_sec.ensure((size_t)nb * 64));
                HIP_TRY(ctx, ctx->hot_mass.ensure((size_t)nb * sizeof(unsigned)));
                HIP_TRY(ctx, ctx->hot_flag.ensure(n_levels));
                auto launch_hot = [&](const unsigned char *flags, unsigned *mass) {
                    hipLaunchKernelGGL(mc::walk_hot_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cum_t.as<double>(),
                                       ctx->block_edge.as<int>(), ctx->ttype.as<int>(), ctx->dest.as<int>(), ctx->tline.as<int>(), flags,
                                       (int)n_levels, (long long)T, (int)S, ctx->hot_sec.as<unsigned>(), mass);
                    return hipGetLastError();
                };
                HIP_TRY(ctx, launch_hot(nullptr, ctx->hot_mass.as<unsigned>()));
                std::vector<unsigned> mass((size_t)nb);
                HIP_TRY(ctx, hipMemcpyAsync(mass.data(), ctx->hot_mass.p, (size_t)nb * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                for (size_t b = 0; b < n_levels; ++b) {
                    const long long rows = o->macro_block_edge_index[b + 1] - o->macro_block_edge_index[b];
                    if (rows <= 0) continue;
                    double m = 0.0;
                    for (size_t sh = 0; sh < S; ++sh) m += (double)mass[sh * n_levels + b];
                    m /= 65536.0 * (double)S;
                    const double need = 1e-3 * (double)(rows > 8 * mc::WALK_WINDOW_QUADS ? ctx->walk_hot_min_mass_long : ctx->walk_hot_min_mass);
                    if (ctx->walk_hot == 1 || m >= need) { hot[b] = 1; ctx->n_hot_blocks += 1; }
                }
                if (ctx->n_hot_blocks > 0) {
                    HIP_TRY(ctx, hipMemcpyAsync(ctx->hot_flag.p, hot.data(), n_levels, hipMemcpyHostToDevice, ctx->stream));
                    hipLaunchKernelGGL(mc::walk_hot_flag_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hot_flag.as<unsigned char>(), nb,
                                       ctx->hot_sec.as<unsigned>());  // (the destinations of the first pass's records, marked; no second walk over the blocks)
                    HIP_TRY(ctx, hipGetLastError());
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    ctx->have_hot = true;
                } else {  // no block qualifies (uniform short blocks): S x levels x 64 B -- 0.5 GB at the configs[4] shape -- are not kept for nothing
                    ctx->hot_sec.release();
                    ctx->hot_flag.release();
                }
                ctx->hot_mass.release();  // (only the choice above read it)
            }
            {
                std::vector<int> bt(2 * n_levels);
                for (size_t b = 0; b < n_levels; ++b) {
                    bt[2 * b] = (int)c0[b];
                    bt[2 * b + 1] = (int)(o->macro_block_edge_index[b + 1] - o->macro_block_edge_index[b]);
                }
                if ((rc = upload(ctx, ctx->blk_tab, bt.data(), bt.size()))) return rc;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            }
            std::vector<int> qi(2 * (size_t)n_quads, 0), lbc(2 * L, 0);  // (quads of the sector padding: {0, 0} -> eight 0xffff entries)
            std::vector<mc::WalkRec> r16((size_t)tc + 1, mc::WalkRec{0u, 0u, 0.0});
            for (size_t b = 0; b < n_levels; ++b) {
                const long long b0 = o->macro_block_edge_index[b], b1 = o->macro_block_edge_index[b + 1];
                for (long long q = c0[b] / 8, k = b0; k < b1; ++q, k += 8) { qi[2 * q] = (int)k; qi[2 * q + 1] = (int)(b1 - k); }
                for (long long k = b0; k < b1; ++k) {
                    const long long c = c0[b] + (k - b0);
                    const int64_t tt = o->transition_type[k];
                    if (tt >= 0) {
                        const int64_t lvl = o->destination_level_id[k];
                        if (hot[lvl]) { r16[c].a = (unsigned)lvl; r16[c].b = mc::WALK_HOT; }
                        else {
                            r16[c].a = (unsigned)c0[lvl];
                            r16[c].b = (unsigned)(o->macro_block_edge_index[lvl + 1] - o->macro_block_edge_index[lvl]);
                        }
                    } else if (tt == -1) {
                        r16[c].a = (unsigned)o->transition_line_id[k];
                        r16[c].b = mc::WALK_EMIT;
                        r16[c].nu = o->line_list_nu[o->transition_line_id[k]];
                    } else
                        r16[c].b = mc::WALK_EMIT | mc::WALK_UNSUPPORTED;
                }
            }
            for (size_t i = 0; i < L; ++i) {
                const int64_t lvl = o->line2macro_level_upper[i];
                if (hot[lvl]) { lbc[2 * i] = (int)lvl; lbc[2 * i + 1] = -1; }
                else {
                    lbc[2 * i] = (int)c0[lvl];
                    lbc[2 * i + 1] = (int)(o->macro_block_edge_index[lvl + 1] - o->macro_block_edge_index[lvl]);
                }
            }
            if ((rc = upload(ctx, ctx->quad_info, qi.data(), qi.size()))) return rc;
            if ((rc = upload(ctx, ctx->rec16, r16.data(), r16.size()))) return rc;
            if ((rc = upload(ctx, ctx->line_block_c, lbc.data(), lbc.size()))) return rc;
            HIP_TRY(ctx, ctx->cum16.ensure((size_t)stride * S * sizeof(unsigned short)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->cum16.p, 0xff, (size_t)stride * S * sizeof(unsigned short), ctx->stream));
            const long long n = n_quads * (long long)S;
            hipLaunchKernelGGL(mc::walk_cum16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cum_t.as<double>(),
                               ctx->quad_info.as<int2>(), n_quads, (long long)T, (int)S, (unsigned)stride, ctx->cum16.as<unsigned short>());
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the host vectors above are the sources of asynchronous copies)
            ctx->cum16_stride = (unsigned)stride;
            ctx->have_walk_tables = true;
        }
    }
    tmark("compact walk tables + hot sectors");
    ctx->lines_sorted = true;
    for (size_t i = 0; i < L; ++i)
        if (!(o->line_list_nu[i] > 0.0) || (i > 0 && !(o->line_list_nu[i] <= o->line_list_nu[i - 1]))) { ctx->lines_sorted = false; break; }
    if (ctx->lines_sorted)
    {   // frequency-bucket index over the (descending) line list.  Resolution: bucket_lines_permille / 1000 lines per bucket on average
        // (default 0.75; rounds 1-4: 3).  A v-packet's shell crossing pins its stopping line with a window of four lines from the
        // bucket's first line: with three lines per bucket 22 % of the crossings missed the window and walked on line by line -- in a wave
        // of ~36 tracing lanes every step of the volley worker loop then waited for such a walk (profiles/r05_bucket_index.txt); at 0.75
        // lines per bucket 1.2 % miss, and the walk takes four lines per round trip (vp_walk_to_stop).  The table is 4 bytes per bucket.
        auto bits = [](double x) { uint64_t u; memcpy(&u, &x, 8); return u; };
        const double nu_hi = o->line_list_nu[0], nu_lo = o->line_list_nu[L - 1];
        int mbits = 4;
        if (nu_lo > 0 && nu_hi >= nu_lo) {
            const double binades = std::max(1.0, std::log2(nu_hi / nu_lo));
            const double per_binade = (double)L / binades;
            while (mbits < 22 && per_binade / (double)(1 << mbits) > 1e-3 * (double)ctx->bucket_lines_permille) ++mbits;
        }
        const int shift = 52 - mbits;
        const long long kmin = (long long)(bits(nu_lo > 0 ? nu_lo : 1.0) >> shift), kmax = (long long)(bits(nu_hi > 0 ? nu_hi : 1.0) >> shift);
        const long long K = std::max<long long>(1, kmax - kmin + 1);
        if (K > (1LL << 26)) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "line list frequency range too wide for the bucket index");
        std::vector<int> first((size_t)K, 0);
        // first[k] = smallest i with key(nu_line[i]) <= kmin + k; keys are non-increasing along the list
        size_t i = 0;
        for (long long k = K - 1; k >= 0; --k) {
            while (i < L && (long long)(bits(o->line_list_nu[i]) >> shift) - kmin > k) ++i;
            first[(size_t)k] = (int)i;
        }
        if ((rc = upload(ctx, ctx->bucket_first, first.data(), first.size()))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->bucket_shift = shift; ctx->bucket_n = (int)K; ctx->bucket_kmin = kmin;
    }
    tmark("sorted check + bucket index");
    if (ctx->have_geometry && (int)S != ctx->n_shells)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "opacity has %zu shells, geometry has %d", S, ctx->n_shells);
    ctx->n_lines = (int)L; ctx->n_trans = (int)T; ctx->n_levels = macro ? (int)E - 1 : 0;
    if (!ctx->have_geometry) ctx->n_shells = (int)S;
    ctx->have_opacity = true;
    return ensure_estimators(ctx);
}

int tardis_mc_set_config(TardisMcContext *ctx, const TardisMcConfig *c)
{
    if (!ctx || !c) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid config");
    if (c->n_spectrum_grid < 0 || (c->n_spectrum_grid > 0 && !c->spectrum_frequency_grid))
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "spectrum_frequency_grid missing");
    if (c->number_of_vpackets > 0 && c->n_spectrum_grid < 2)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "v-packets need a spectrum grid with >= 2 edges");
    if (c->line_interaction_type < 0 || c->line_interaction_type > 2)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "line_interaction_type must be 0, 1 or 2");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->cfg = *c;
    ctx->grid_host.assign(c->spectrum_frequency_grid, c->spectrum_frequency_grid + c->n_spectrum_grid);
    ctx->cfg.spectrum_frequency_grid = ctx->grid_host.data();
    int rc;
    if ((rc = upload(ctx, ctx->grid, ctx->grid_host.data(), ctx->grid_host.size()))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_config = true;
    return ensure_estimators(ctx);
}

int tardis_mc_set_packets(TardisMcContext *ctx, const TardisMcPackets *p)
{
    if (!ctx || !p || p->n_packets < 0 ||
        (p->n_packets > 0 && (!p->initial_radii || !p->initial_nus || !p->initial_mus || !p->initial_energies || !p->packet_seeds)))
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid packets");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)p->n_packets;
    int rc;
    HIP_TRY(ctx, ctx->r0.ensure(P * 8)); HIP_TRY(ctx, ctx->mu0.ensure(P * 8)); HIP_TRY(ctx, ctx->nu0.ensure(P * 8)); HIP_TRY(ctx, ctx->e0.ensure(P * 8));
    HIP_TRY(ctx, ctx->seeds.ensure(P * 4));
    // The seeds arrive as int64 and the kernels read uint32 (np.random.seed(int) -> init_genrand(uint32)): sent as they are and narrowed on the device.  (A host
    // vector of P words -- zero-filled, then filled -- cost 100 ms of the 200 ms this call took at 1e8 packets; the extra 4 bytes per packet over the link cost 8.)
    HIP_TRY(ctx, ctx->staging.ensure(P * 8));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (nothing of an earlier call still reads the packet buffers)
    HIP_TRY(ctx, host_copy(ctx, {{(void *)p->initial_radii, ctx->r0.p, P * 8}, {(void *)p->initial_mus, ctx->mu0.p, P * 8},
                                      {(void *)p->initial_nus, ctx->nu0.p, P * 8}, {(void *)p->initial_energies, ctx->e0.p, P * 8},
                                      {(void *)p->packet_seeds, ctx->staging.p, P * 8}}, true));
    if (P > 0) {
        hipLaunchKernelGGL(narrow_seeds_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream, ctx->staging.as<long long>(), (long long)P, ctx->seeds.as<uint32_t>());
        HIP_TRY(ctx, hipGetLastError());
    }
    (void)rc;
    HIP_TRY(ctx, ctx->out_nu.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->out_e.ensure(P * sizeof(double)));
    if (ctx->track) {
        for (auto &b : ctx->li_f64) HIP_TRY(ctx, b.ensure(P * sizeof(double)));
        for (auto &b : ctx->li_i64) HIP_TRY(ctx, b.ensure(P * sizeof(long long)));
        HIP_TRY(ctx, ctx->li_rec.ensure(P * 64));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_packets = (long long)P;
    ctx->have_packets = true;
    return TARDIS_MC_OK;
}

/* ---- black-body packet source on the device ------------------------------------------------------- */
namespace {
typedef unsigned __int128 hu128;
inline hu128 h128(uint64_t hi, uint64_t lo) { return ((hu128)hi << 64) | lo; }
const hu128 PCG_MULT = h128(2549297995355413924ULL, 4865540595714422341ULL);  // PCG_DEFAULT_MULTIPLIER_128

mc::PcgAffine affine(hu128 mult, hu128 plus)
{
    return mc::PcgAffine{(uint64_t)mult, (uint64_t)(mult >> 64), (uint64_t)plus, (uint64_t)(plus >> 64)};
}
}  // namespace

int tardis_mc_pcg64_seed(uint64_t seed, uint64_t out_state[4])
{
    if (!out_state) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    // numpy.random.SeedSequence(seed) with an empty spawn key, pool of 4 words (numpy/random/bit_generator.pyx)
    const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu,
                   MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t entropy[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_entropy = entropy[1] ? 2 : 1;
    uint32_t hash_const = INIT_A;
    auto hashmix = [&](uint32_t v) {
        v ^= hash_const; hash_const *= MULT_A; v *= hash_const; v ^= v >> 16; return v;
    };
    auto mix = [&](uint32_t x, uint32_t y) { uint32_t r = MIX_L * x - MIX_R * y; r ^= r >> 16; return r; };
    uint32_t pool[4];
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_entropy ? entropy[i] : 0u);
    for (int i_src = 0; i_src < 4; ++i_src)
        for (int i_dst = 0; i_dst < 4; ++i_dst)
            if (i_src != i_dst) pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src]));
    // generate_state(4, uint64): 8 words, little-endian pairs
    uint32_t words[8];
    uint32_t hc = INIT_B;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hc; hc *= MULT_B; v *= hc; v ^= v >> 16;
        words[i] = v;
    }
    uint64_t st[4];
    for (int i = 0; i < 4; ++i) st[i] = (uint64_t)words[2 * i] | ((uint64_t)words[2 * i + 1] << 32);
    // pcg64_set_seed: initstate = (st[0] high, st[1] low), initseq = (st[2] high, st[3] low); pcg_setseq_128_srandom_r
    const hu128 initstate = h128(st[0], st[1]), initseq = h128(st[2], st[3]);
    const hu128 inc = (initseq << 1) | 1;
    hu128 state = 0;
    state = state * PCG_MULT + inc;
    state += initstate;
    state = state * PCG_MULT + inc;
    out_state[0] = (uint64_t)(state >> 64); out_state[1] = (uint64_t)state;
    out_state[2] = (uint64_t)(inc >> 64); out_state[3] = (uint64_t)inc;
    return TARDIS_MC_OK;
}

static int ensure_packet_buffers(TardisMcContext *ctx, size_t P)
{
    HIP_TRY(ctx, ctx->r0.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->mu0.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->nu0.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->e0.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->seeds.ensure(P * sizeof(uint32_t)));
    HIP_TRY(ctx, ctx->out_nu.ensure(P * sizeof(double)));
    HIP_TRY(ctx, ctx->out_e.ensure(P * sizeof(double)));
    if (ctx->track) {
        for (auto &b : ctx->li_f64) HIP_TRY(ctx, b.ensure(P * sizeof(double)));
        for (auto &b : ctx->li_i64) HIP_TRY(ctx, b.ensure(P * sizeof(long long)));
        HIP_TRY(ctx, ctx->li_rec.ensure(P * 64));
    }
    return TARDIS_MC_OK;
}

int tardis_mc_create_blackbody_packets(TardisMcContext *ctx, int64_t n_total, int64_t first, int64_t count, double radius,
                                       double temperature, const uint64_t pcg_state[4], uint32_t max_seed_val,
                                       const double *l_array, int64_t n_l)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (n_total < 0 || first < 0 || count < 0 || first + count > n_total || !pcg_state || !l_array || n_l < 1 || n_l > (1 << 24) ||
        max_seed_val < 2 || (uint64_t)n_total >= (1ULL << 44))
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid packet source arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)count;
    int rc = ensure_packet_buffers(ctx, P);
    if (rc) return rc;
    // jump tables of the LCG  s -> M s + inc
    const hu128 inc = h128(pcg_state[2], pcg_state[3]);
    std::vector<mc::PcgAffine> jump(mc::PCG_JUMP_BITS);
    hu128 jm[mc::PCG_JUMP_BITS], jp[mc::PCG_JUMP_BITS];
    jm[0] = PCG_MULT; jp[0] = inc;
    for (int j = 1; j < mc::PCG_JUMP_BITS; ++j) { jm[j] = jm[j - 1] * jm[j - 1]; jp[j] = (jm[j - 1] + 1) * jp[j - 1]; }
    for (int j = 0; j < mc::PCG_JUMP_BITS; ++j) jump[j] = affine(jm[j], jp[j]);
    hu128 nm = 1, np_ = 0;  // n_total-step map, composed from the set bits
    for (int j = 0; j < mc::PCG_JUMP_BITS; ++j)
        if (((uint64_t)n_total >> j) & 1) { nm = jm[j] * nm; np_ = jm[j] * np_ + jp[j]; }
    DevBuf d_jump, d_l, d_rej, d_cnt;
    HIP_TRY(ctx, d_jump.ensure(jump.size() * sizeof(mc::PcgAffine)));
    HIP_TRY(ctx, hipMemcpyAsync(d_jump.p, jump.data(), jump.size() * sizeof(mc::PcgAffine), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, d_l.ensure((size_t)n_l * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(d_l.p, l_array, (size_t)n_l * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int capacity = 1 << 16;
    HIP_TRY(ctx, d_rej.ensure((size_t)capacity * sizeof(long long)));
    HIP_TRY(ctx, d_cnt.ensure(sizeof(unsigned int)));
    HIP_TRY(ctx, hipMemsetAsync(d_cnt.p, 0, sizeof(unsigned int), ctx->stream));

    mc::PacketSourceArgs a{};
    a.n_total = n_total; a.first = first; a.count = count;
    a.state_hi = pcg_state[0]; a.state_lo = pcg_state[1];
    a.jump = d_jump.as<mc::PcgAffine>();
    a.step_n = affine(nm, np_);
    a.seed_range_excl = max_seed_val;
    a.seed_threshold = (uint32_t)((0x100000000ULL - max_seed_val) % max_seed_val);  // (UINT32_MAX - rng) % rng_excl
    a.l_array = d_l.as<double>(); a.n_l = (int)n_l;
    a.l_coef = 1.082323233711138;  // np.pi**4 / 90.0 (black_body.py:175)
    a.radius = radius;
    a.kT = 1.3806488e-16 * temperature;   // tardis/constants.py:1 (CODATA 2010, cgs)
    a.h = 6.62606957e-27;
    a.energy = n_total > 0 ? 1.0 / (double)n_total : 0.0;
    a.r0 = ctx->r0.as<double>(); a.mu0 = ctx->mu0.as<double>(); a.nu0 = ctx->nu0.as<double>(); a.e0 = ctx->e0.as<double>();
    a.seeds = ctx->seeds.as<uint32_t>();

    // pass 1: rejected u32 positions of the bounded-integer draw (threshold / 2^32 of all draws; none for almost every
    // run with the reference's MAX_SEED_VAL = 2^32 - 1)
    std::vector<long long> rejected;
    long long consumed_u32 = 0;
    if (n_total > 0) {
        unsigned int n_rej = 0;
        if (a.seed_threshold > 0) {
            const long long n_positions = n_total + capacity;
            const int blocks = (int)std::min<long long>((n_positions / 2 + 255) / 256 + 1, 8192);
            hipLaunchKernelGGL(mc::packet_source_scan_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a, n_positions,
                               d_rej.as<long long>(), capacity, d_cnt.as<unsigned int>());
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(&n_rej, d_cnt.p, sizeof(n_rej), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if ((long long)n_rej >= capacity)
                return fail(ctx, TARDIS_MC_ERR_UNSUPPORTED, "packet source: too many rejected seed draws for this seed range");
            rejected.resize(n_rej);
            if (n_rej) {
                HIP_TRY(ctx, hipMemcpy(rejected.data(), d_rej.p, n_rej * sizeof(long long), hipMemcpyDeviceToHost));
                std::sort(rejected.begin(), rejected.end());
                HIP_TRY(ctx, hipMemcpy(d_rej.p, rejected.data(), n_rej * sizeof(long long), hipMemcpyHostToDevice));
            }
        }
        long long pos = n_total - 1;
        for (size_t k = 0; k < rejected.size() && rejected[k] <= pos; ++k) ++pos;
        consumed_u32 = pos + 1;
    }
    a.rejected = d_rej.as<long long>();
    a.n_rejected = (int)rejected.size();
    a.xi_first_u64 = (consumed_u32 + 1) / 2;
    if (count > 0) {
        const int blocks = (int)std::min<long long>((count + 255) / 256, 16384);
        hipLaunchKernelGGL(mc::packet_source_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    d_jump.release(); d_l.release(); d_rej.release(); d_cnt.release();
    ctx->n_packets = count;
    ctx->have_packets = true;
    return TARDIS_MC_OK;
}

int tardis_mc_get_packets(TardisMcContext *ctx, double *initial_radii, double *initial_nus, double *initial_mus,
                          double *initial_energies, int64_t *packet_seeds)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->have_packets) return fail(ctx, TARDIS_MC_ERR_STATE, "no packets resident");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_packets;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (initial_radii && P) HIP_TRY(ctx, hipMemcpy(initial_radii, ctx->r0.p, P * 8, hipMemcpyDeviceToHost));
    if (initial_nus && P) HIP_TRY(ctx, hipMemcpy(initial_nus, ctx->nu0.p, P * 8, hipMemcpyDeviceToHost));
    if (initial_mus && P) HIP_TRY(ctx, hipMemcpy(initial_mus, ctx->mu0.p, P * 8, hipMemcpyDeviceToHost));
    if (initial_energies && P) HIP_TRY(ctx, hipMemcpy(initial_energies, ctx->e0.p, P * 8, hipMemcpyDeviceToHost));
    if (packet_seeds && P) {
        std::vector<uint32_t> tmp(P);
        HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->seeds.p, P * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < P; ++i) packet_seeds[i] = (int64_t)tmp[i];
    }
    return TARDIS_MC_OK;
}

int tardis_mc_reset_estimators(TardisMcContext *ctx)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_estimators(ctx);
    if (rc) return rc;
    if (!ctx->est_valid) return fail(ctx, TARDIS_MC_ERR_STATE, "set_opacity and set_config must precede reset_estimators");
    EstLayout e = est_layout(ctx->est_S, ctx->est_L, ctx->est_G, ctx->est_copies);
    HIP_TRY(ctx, hipMemsetAsync(ctx->est.p, 0, e.total * sizeof(double), ctx->stream));
    HIP_TRY(ctx, ctx->counters.ensure(TARDIS_MC_N_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, TARDIS_MC_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    return TARDIS_MC_OK;
}

int tardis_mc_propagate(TardisMcContext *ctx)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->have_geometry || !ctx->have_opacity || !ctx->have_config || !ctx->have_packets)
        return fail(ctx, TARDIS_MC_ERR_STATE, "set_geometry/set_opacity/set_config/set_packets must precede propagate");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->compactions = 0;
    const bool rs_armed = ctx->rs.armed;  // (tardis_mc_stream_results holds for one call)
    ctx->rs.armed = false; ctx->rs.valid = false; ctx->rs.upto = 0; ctx->rs.n_late = 0;
    const int tune_pending = ctx->ls_tune.pending;  // (the lane-sweep tuner: whether the previous propagate call was one of its timed ones: 2 * instantiation + sample)
    ctx->ls_tune.pending = -1;
    int tune_slot = -1;  // this call is a timed one of the tuner (becomes ls_tune.pending once it has been enqueued completely)
    int rc = ensure_estimators(ctx);
    if (rc) return rc;
    if (!ctx->counters.p) {
        HIP_TRY(ctx, ctx->counters.ensure(TARDIS_MC_N_COUNTERS * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, TARDIS_MC_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    }
    const TardisMcConfig &c = ctx->cfg;
    const bool vpk = c.number_of_vpackets > 0;
    if (ctx->track) {  // (tracking may have been switched on after the packets were set)
        const size_t P = (size_t)std::max<long long>(ctx->n_packets, 1);
        for (auto &b : ctx->li_f64) HIP_TRY(ctx, b.ensure(P * sizeof(double)));
        for (auto &b : ctx->li_i64) HIP_TRY(ctx, b.ensure(P * sizeof(long long)));
        HIP_TRY(ctx, ctx->li_rec.ensure(P * 64));
    }
    // v-packet log buffers
    if (c.enable_vpacket_tracking && vpk) {
        // (sized for the current call: the engine is cached per process, a later, larger run must not inherit a smaller log)
        if (!ctx->vlog_capacity_user) ctx->vlog_capacity = std::max<long long>(1024, ctx->n_packets * c.number_of_vpackets * 64);
        size_t cap = (size_t)ctx->vlog_capacity;
        HIP_TRY(ctx, ctx->vlog_count.ensure(sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->vlog_count.p, 0, sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, ctx->vlog_packet.ensure(cap * sizeof(long long)));
        HIP_TRY(ctx, ctx->vlog_seq.ensure(cap * sizeof(int)));
        HIP_TRY(ctx, ctx->vlog_nu.ensure(cap * sizeof(double)));
        HIP_TRY(ctx, ctx->vlog_energy.ensure(cap * sizeof(double)));
        HIP_TRY(ctx, ctx->vlog_mu.ensure(cap * sizeof(double)));
        HIP_TRY(ctx, ctx->vlog_r.ensure(cap * sizeof(double)));
    }
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    HIP_TRY(ctx, ctx->first_error.ensure(2 * sizeof(long long)));
    // (argument blocks reach the device through store_value(): consecutive propagate calls -- iterations, chunks submitted by
    // the host -- are not serialised by a stream synchronisation here)
    HIP_TRY(ctx, store_value(ctx->stream, ctx->first_error.as<FirstErrorInit>(), FirstErrorInit{{0x7fffffffffffffffLL, 0}}));
    ctx->progress_done = false; ctx->progress_wave = false; ctx->progress_total = ctx->n_packets;
    {
        std::lock_guard<std::mutex> lock(ctx->progress_mutex);  // (tardis_mc_progress may be reading next_packet)
        HIP_TRY(ctx, ctx->next_packet.ensure(sizeof(unsigned long long)));
        if (!ctx->ev_progress_reset) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_progress_reset, hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->next_packet.p, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_progress_reset, ctx->stream));
    // the cooperative kernel relies on a sorted line list (bucket index, monotone stopping predicate); anything else --
    // which the reference would also mis-handle -- goes through the sequential lane-per-packet kernel
    // automatic choice: the wave-owner kernel (its pooled v-packet volleys take up to 32 v-packets per volley: one bit of
    // the roulette predictor each; beyond that the lane-per-packet kernel); lane sweeps where their bounds hold (partial
    // relativity) and no volleys run.  On the macroatom shape (5e5 lines, ~36 lines per trace) the lane sweeps overtook the
    // group sweeps once the walk ran per lane on the compact tables and a call became epochs over one packet supply
    // (19.3 vs 13.4 Mpkt/s at 2e7 packets): the group sweeps' 280 instructions per 16-line step had become the bound.
    // v-packet screening (tau_prefix.hpp): with the default survival probability 0 a v-packet whose optical depth passes
    // tau_russian is dropped whatever the depth was -- decided from prefix sums, two reads per shell crossing.
    // It pays where a shell crossing passes many lines (two prefix reads against ~40 optical depths on the 100-shell x 5e5-line
    // shape: 1.7x - 2.1x); on the tardis_example shape (~12 lines per crossing, most v-packets leave the grid alive) the
    // screening is a second trace on top of the first: -36 % (profiles/r03_vpacket_screening.txt).  "vpacket_screening" 0 / 1
    // overrides the automatic choice.
    // (decided from cheap predicates first: whether the tables are BUILT depends on the kernel the call ends up on -- calls that
    // take the lane kernel (more than 32 v-packets per volley, unsorted line list, variant 0) never read them: S x (L + 1)
    // doubles, 400 MB at the configs[4] shape, and a blocking read-back of the negative-depth flag)
    bool screen_on = false;
    {
        const bool screen_auto = (long long)ctx->n_lines >= 2500LL * (long long)ctx->n_shells;
        const bool screen = ctx->vpacket_screening < 0 ? screen_auto : ctx->vpacket_screening != 0;
        screen_on = vpk && c.survival_probability == 0.0 && screen && !(ctx->debug_flags & 33554432) &&
                    !(ctx->pfx_valid && ctx->pfx_negative);
    }
    auto build_screening_tables = [&]() -> int {  // first v-packet call after set_opacity that screens
        if (ctx->pfx_valid) return TARDIS_MC_OK;
        const size_t S = (size_t)ctx->n_shells, L = (size_t)ctx->n_lines;
        HIP_TRY(ctx, ctx->tau_pfx.ensure((S * (L + 1) + 8) * sizeof(double)));  // (+8: the four-entry windows of the screening)
        HIP_TRY(ctx, ctx->tau_rowsum.ensure(S * sizeof(double)));
        HIP_TRY(ctx, ctx->pfx_flag.ensure(sizeof(int)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->pfx_flag.p, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(mc::tau_prefix_kernel, dim3((unsigned)S), dim3(256), 0, ctx->stream, ctx->tau_t.as<double>(), (int)L,
                           ctx->tau_pfx.as<double>(), ctx->tau_rowsum.as<double>(), ctx->pfx_flag.as<int>());
        HIP_TRY(ctx, hipGetLastError());
        int neg = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&neg, ctx->pfx_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pfx_negative = neg != 0;
        ctx->pfx_valid = true;
        return TARDIS_MC_OK;
    };
    const bool prefer_lane_sweeps = !c.enable_full_relativity && !vpk;
    // With v-packets the wave kernel's pooled volleys win where a v-packet crosses few shells and few lines (the tardis_example
    // shape: 8.0 vs 5.2 Mpkt/s); on finer grids and longer line lists the group kernel -- every lane of a packet's group traces one
    // v-packet of the volley, no speculation on the draw positions -- was measured 1.2x to 2.2x ahead
    // (profiles/r02_vpacket_kernel_choice.txt).
    // With the screening the v-packets of such a shape are half of the wave kernel's pass instead of nearly all of it, and its
    // lane-per-packet event code wins once the call is long enough to amortise its drain: 1.39-1.46 vs 1.33 Mpkt/s at 3e6 packets of
    // the configs[4] shape, 0.63 vs 1.10 at 1e6 (profiles/r03_vpacket_screening.txt).
    // (round 5: with the finer bucket index, the carried walks and the cut-off of the volley phases the wave kernel is ahead from 1e6 packets per call on:
    // 1.20 vs 1.81 s there, 4.5 vs 12.7 s at 1e7 -- and still at 1e5, 0.63 vs 0.74 s: profiles/r05_vpacket_kernel_choice.txt; the threshold was 2.5e6 in round 3)
    const bool vpk_wave = vpk && ((ctx->n_shells <= 30 && ctx->n_lines <= 100000) || (screen_on && ctx->n_packets >= ctx->vpk_wave_min_packets));
    int variant = ctx->variant >= 0 ? ctx->variant
                                    : ((vpk && c.number_of_vpackets > 32) ? 0 : (vpk ? (vpk_wave ? 2 : 1) : (prefer_lane_sweeps ? 3 : 2)));
    if (ctx->prob_negative && (variant == 2 || variant == 3)) variant = 1;  // (the wave kernel searches the monotone running sums)
    // Russian roulette with survivors (virtual_packet.py:221-232; the reference's SURVIVAL_PROBABILITY is 0 in every run, nothing
    // sets it): a surviving v-packet may play again in a later shell, so its draw count is unbounded, while the wave kernel's
    // pooled volleys budget one roulette draw per v-packet -- such problems run on the group kernel
    if (vpk && c.survival_probability > 0.0 && (variant == 2 || variant == 3 || variant == 4)) variant = 1;
    // variant 4: the wave kernel with the volley queue (v-packets traced by vpacket_trace_kernel between its launches); without
    // v-packets there is nothing to queue
    if (variant == 4 && !vpk) variant = prefer_lane_sweeps ? 3 : 2;
    if (ctx->prob_negative && variant == 4) variant = 1;
    const bool cooperative = ctx->lines_sorted && (variant == 1 || variant == 2 || variant == 3 || variant == 4) && (!vpk || c.number_of_vpackets <= 32);
    ctx->last_variant = cooperative ? ((variant == 3 && c.enable_full_relativity) ? 2 : variant) : 0;
    if (screen_on && !cooperative) screen_on = false;  // (the lane kernel traces line by line)
    if (screen_on) {
        rc = build_screening_tables();
        if (rc) return rc;
        if (ctx->pfx_negative) {
            // a negative optical depth: no screening (the prefix sums would not bound the serial sum).  The automatic kernel
            // choice counted on it for long calls of the wave kernel: take what it picks without the screening.
            screen_on = false;
            if (ctx->variant < 0 && variant == 2 && vpk && !(ctx->n_shells <= 30 && ctx->n_lines <= 100000)) {
                variant = 1;
                ctx->last_variant = variant;
            }
        }
    }

    if (!cooperative) {
        // variant 0: lane-per-packet, persistent-ish grid, static round-robin packet assignment
        // (the context is cached: the timing queries must not report the epochs of an earlier wave-kernel call)
        ctx->wave_epoch_mode = false;
        ctx->post_pending[0] = ctx->post_pending[1] = false;
        ctx->prop_pending = false;
        ctx->sum_seed_ms = ctx->sum_prop_ms = ctx->sum_post_ms = 0.0;
        ctx->launches = 0;
        long long want_blocks = (ctx->n_packets + 255) / 256;
        int blocks = (int)std::max<long long>(1, std::min<long long>(want_blocks, (long long)cus * ctx->blocks_per_cu));
        HIP_TRY(ctx, ctx->rng_state.ensure((size_t)blocks * 256 * mc::MT_N * sizeof(uint32_t)));
        mc::DeviceProblem P = make_device_problem(ctx);
        const size_t lds = 2 * (size_t)ctx->n_shells * sizeof(double);
        if (lds > 64 * 1024) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "n_shells too large for the LDS J/nu_bar accumulator");
        HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        ctx->chunks_timed = 0;
        if (ctx->n_packets > 0) {
            if (c.enable_full_relativity) { if (vpk) launch_lane<true, true>(ctx, P, blocks, lds); else launch_lane<true, false>(ctx, P, blocks, lds); }
            else { if (vpk) launch_lane<false, true>(ctx, P, blocks, lds); else launch_lane<false, false>(ctx, P, blocks, lds); }
            HIP_TRY(ctx, hipGetLastError());
        }
    } else {
        // variant 1: group kernel, chunked (MT19937 states are seeded per chunk by a lane-per-packet kernel);
        // variants 2 / 3: wave-owner kernel, one packet supply for the whole call, launched in epochs (see LaneSave)
        const bool wave_kernel = variant == 2 || variant == 3 || variant == 4;
        const bool vq = variant == 4;
        ctx->problem_host = make_device_problem(ctx);
        const mc::DeviceProblem &F = ctx->problem_host;
        HIP_TRY(ctx, ctx->problem_dev.ensure(sizeof(mc::DeviceProblem)));
        HIP_TRY(ctx, store_value(ctx->stream, ctx->problem_dev.as<mc::DeviceProblem>(), ctx->problem_host));
        mc::GroupArgs P{};
        P.cold = ctx->problem_dev.as<mc::DeviceProblem>();
        P.n_shells = F.n_shells; P.n_lines = F.n_lines; P.n_trans = F.n_trans;
        P.line_interaction_type = F.line_interaction_type; P.disable_line_scattering = F.disable_line_scattering;
        P.debug_flags = F.debug_flags; P.n_est_copies = F.n_est_copies;
        P.t_exp = F.t_exp; P.sigma_thomson = F.sigma_thomson;
        P.tc = F.t_exp * mc::C_LIGHT; P.rcp_tc = 1.0 / P.tc;
        if ((long long)ctx->n_shells * ctx->n_lines >= (1LL << 28) || (long long)ctx->n_shells * ctx->n_trans >= (1LL << 28))
            return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "n_shells * n_lines exceeds the 32-bit table offsets of the cooperative kernel");
        if (wave_kernel && ctx->n_packets >= (1LL << 31))
            return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "more than 2^31 packets per propagate call");
        P.r_inner = F.r_inner; P.r_outer = F.r_outer; P.nu_line = F.nu_line; P.tau_t = F.tau_t; P.n_e = F.n_e; P.prob_t = F.prob_t;
        P.line_block = ctx->line_block.as<int2>(); P.trans_rec = ctx->trans_rec.as<int4>();
        P.cum_t = ctx->cum_t.as<double>(); P.trans_nu = ctx->trans_nu.as<double>();
        P.jblue_t = F.jblue_t; P.edot_t = F.edot_t; P.est_copy_stride = F.est_copy_stride;
        P.next_packet = F.next_packet;
        P.n_vpackets = F.n_vpackets; P.survival_probability = F.survival_probability; P.tau_russian = F.tau_russian;
        P.spawn_start = F.spawn_start; P.spawn_end = F.spawn_end; P.grid0 = F.grid0; P.grid_last = F.grid_last;
        P.delta_nu = F.delta_nu; P.vhist = F.vhist;
        P.bucket_first = ctx->bucket_first.as<int>(); P.bucket_shift = ctx->bucket_shift; P.bucket_n = ctx->bucket_n;
        P.bucket_kmin = ctx->bucket_kmin;
        P.tau_pfx = screen_on ? ctx->tau_pfx.as<double>() : nullptr;
        P.tau_rowsum = screen_on ? ctx->tau_rowsum.as<double>() : nullptr;
        // macro-atom jumps of the wave kernel (macroatom chains and the single jump of downbranch alike): per-lane walk on the
        // compact tables (walk_tables.hpp); debug flag 8192 keeps the cooperative group scan of the fp64 running sums (macroatom) /
        // the fp64 search (downbranch), 128 the per-lane search in them (both for cross-checks)
        const bool compact_walk = wave_kernel && c.line_interaction_type != 0 && ctx->have_walk_tables && !(ctx->debug_flags & (128 | 8192));
        if (compact_walk) {
            P.cum16 = ctx->cum16.as<unsigned short>(); P.rec16 = ctx->rec16.as<mc::WalkRec>(); P.quad_info = ctx->quad_info.as<int2>();
            P.cum16_stride = ctx->cum16_stride;
            P.line_block = ctx->line_block_c.as<int2>();
            P.hot_sec = ctx->have_hot ? ctx->hot_sec.as<unsigned>() : nullptr;
            P.blk_tab = ctx->blk_tab.as<int2>();
            P.hot_stride = (unsigned)(16u * (unsigned)ctx->n_levels);
        }
        const bool full = c.enable_full_relativity != 0, trk = ctx->track;
        while ((int)ctx->ev_chunk.size() < 8) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev_chunk.push_back(e);
        }
        ctx->chunks_timed = 0;
        ctx->sum_seed_ms = ctx->sum_prop_ms = ctx->sum_post_ms = 0.0;
        ctx->launches = 0;
        if (wave_kernel) {
            const bool lane_sweep = variant == 3 && !full;  // (the bounds of the lane sweep are those of partial relativity)
            size_t wave_lds = lane_sweep ? (vpk ? mc::wave_kernel_lds_bytes<false, true, true>(ctx->n_shells) : mc::wave_kernel_lds_bytes<false, false, true>(ctx->n_shells))
                                  : vpk ? (full ? mc::wave_kernel_lds_bytes<true, true>(ctx->n_shells) : mc::wave_kernel_lds_bytes<false, true>(ctx->n_shells))
                                        : (full ? mc::wave_kernel_lds_bytes<true, false>(ctx->n_shells) : mc::wave_kernel_lds_bytes<false, false>(ctx->n_shells));
            if (wave_lds > 64 * 1024) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "n_shells too large for the LDS J/nu_bar accumulator");
            const int wave_waves_per_cu = std::max(1, std::min(ctx->waves_per_simd > 0 ? 4 * ctx->waves_per_simd : 16, (int)((160 * 1024) / wave_lds)));
            using WaveKernelFn = void (*)(mc::WaveHot, const mc::WaveCold *);
            WaveKernelFn kw = nullptr;
            // (XW: the instantiations with the macro-atom walks on the fp64 running sums compiled in -- only launched when the compact
            // walk tables are not used: debug flags 128 / 8192, tables too large; the production ones are 22 % shorter without them)
#define TMC_PICKW3(G_, V_, X_) (full ? (trk ? mc::propagate_wave_kernel<true, true, G_, V_, false, X_> : mc::propagate_wave_kernel<true, false, G_, V_, false, X_>) \
                                     : (trk ? mc::propagate_wave_kernel<false, true, G_, V_, false, X_> : mc::propagate_wave_kernel<false, false, G_, V_, false, X_>))
#define TMC_PICKW2(G_, V_) (xwalk ? TMC_PICKW3(G_, V_, true) : TMC_PICKW3(G_, V_, false))
#define TMC_PICKW(G_) (vpk ? TMC_PICKW2(G_, true) : TMC_PICKW2(G_, false))
#define TMC_PICKLS2(V_, X_) (trk ? mc::propagate_wave_kernel<false, true, 16, V_, true, X_> : mc::propagate_wave_kernel<false, false, 16, V_, true, X_>)
#define TMC_PICKLS(V_) (xwalk ? TMC_PICKLS2(V_, true) : TMC_PICKLS2(V_, false))
            // (flag 1048576: the long instantiations, for A/B; the flags that read the kernel's profiling / test counters: those are only compiled into the long ones)
            const int dbg_counter_flags = mc::WV_DBG_FLAGS;  // (defined next to the kernel's DBG-gated code: propagate_wave.hpp)
            const bool xwalk = (c.line_interaction_type != 0 && !compact_walk) || (ctx->debug_flags & (1048576 | dbg_counter_flags)) != 0;
            // (sweep-worker width of the wave kernel: 8 lanes for sparse line lists, 16 for long ones, like the group kernel; the lane-sweep
            // instantiations only use it in the cross-check walks: one width)
            const int GW = ctx->group_size ? ctx->group_size : (ctx->n_lines <= 100000 ? 8 : 16);
            if (lane_sweep) kw = vpk ? TMC_PICKLS(true) : TMC_PICKLS(false);
            else kw = (GW == 16) ? TMC_PICKW(16) : (GW == 4 ? TMC_PICKW(4) : TMC_PICKW(8));
            // v-packets on a grid so fine that the per-shell LDS arrays leave room for at most eight waves per CU (two per SIMD): the
            // instantiation compiled for two waves per SIMD -- 239 VGPRs, no spills -- costs no occupancy there (built for the sweep widths G = 16 and
            // G = 8, without the cross-check walks; option vpk_wide_registers 0 keeps the 168-VGPR one)
            // Measured (profiles/r05_vpk_wide_registers.txt): 3727-3766 vs 4442-4450 ms per 1e7 packets of the configs[4] shape (-16 %).  Option 2 forces
            // it (then eight waves per CU whatever the LDS allows), 0 keeps the 168-VGPR instantiation.
            bool wide = vpk && !xwalk && (lane_sweep || GW == 16 || GW == 8) &&
                        ((ctx->vpk_wide_registers == 1 && wave_waves_per_cu <= 8) || ctx->vpk_wide_registers == 2);
#define TMC_PICKWIDE(G_) (full ? (trk ? mc::propagate_wave_kernel<true, true, G_, true, false, false, 2> : mc::propagate_wave_kernel<true, false, G_, true, false, false, 2>) \
                               : (trk ? mc::propagate_wave_kernel<false, true, G_, true, false, false, 2> : mc::propagate_wave_kernel<false, false, G_, true, false, false, 2>))
            if (wide) kw = lane_sweep ? (trk ? mc::propagate_wave_kernel<false, true, 16, true, true, false, 2> : mc::propagate_wave_kernel<false, false, 16, true, true, false, 2>)
                                      : (GW == 16 ? TMC_PICKWIDE(16) : TMC_PICKWIDE(8));  // (lane sweeps with v-packets: variant 3 under partial relativity)
#undef TMC_PICKWIDE
            // which lane-sweep instantiation (see ls_waves_per_simd above): forced by the option, or timed on the first calls of this key
            bool ls3 = false;
            if (lane_sweep && !vpk && !xwalk) {
                auto &tn = ctx->ls_tune;
                // (a call that does not even fill the grid's lanes four times over is nothing but the drain of its longest packets: B, measured -6 % on
                // 1e5 - 1e6-packet calls of the tardis_example shape, without spending five calls of a 20-iteration run on finding that out)
                const bool all_drain = ctx->n_packets < 4LL * 64 * 16 * cus;
                if (ctx->ls_waves_per_simd == 3 || (ctx->ls_waves_per_simd == 0 && all_drain)) ls3 = true;
                else if (ctx->ls_waves_per_simd == 0 && ctx->pass_cus == 0) {
                    if (tn.n != ctx->n_packets || tn.lines != ctx->n_lines || tn.shells != ctx->n_shells || tn.mode != c.line_interaction_type ||
                        tn.table != ctx->sweep_table) {
                        tn.n = ctx->n_packets; tn.lines = ctx->n_lines; tn.shells = ctx->n_shells; tn.mode = c.line_interaction_type; tn.table = ctx->sweep_table;
                        tn.phase = 0; tn.choice = 0;
                        tn.ms[0][0] = tn.ms[0][1] = tn.ms[1][0] = tn.ms[1][1] = -1.0;
                    } else if (tune_pending >= 0) {  // the previous call of this key was a timed one: its duration (propagation + passes)
                        float ms = 0.f;
                        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_tune[1]));
                        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_tune[0], ctx->ev_tune[1]));
                        tn.ms[tune_pending >> 1][tune_pending & 1] = ms;
                    } else if (tn.phase >= 2 && tn.phase <= 5)
                        tn.phase -= 1;  // the previous phase was a timed call and left no measurement (the call failed half-way): again
                    // call 0: A, not timed (first-call allocations, the log sized from a guess); calls 1 / 3: A; calls 2 / 4: B; from call 5 on: the choice
                    if (tn.phase == 0) ls3 = false;
                    else if (tn.phase <= 4) { ls3 = (tn.phase & 1) == 0; tune_slot = 2 * (ls3 ? 1 : 0) + ((tn.phase - 1) >> 1); }
                    else {
                        if (tn.phase == 5) {
                            const bool all = tn.ms[0][0] > 0.0 && tn.ms[0][1] > 0.0 && tn.ms[1][0] > 0.0 && tn.ms[1][1] > 0.0;
                            tn.choice = (all && std::min(tn.ms[1][0], tn.ms[1][1]) < 0.97 * std::min(tn.ms[0][0], tn.ms[0][1])) ? 1 : 0;
                        }
                        ls3 = tn.choice == 1;
                    }
                    if (tn.phase < 6) ++tn.phase;
                    if (tune_slot >= 0)
                        for (int k = 0; k < 2; ++k)
                            if (!ctx->ev_tune[k]) HIP_TRY(ctx, hipEventCreate(&ctx->ev_tune[k]));
                }
            }
            if (ls3) kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 3> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 3>;
            // the interleaved sweep table (option sweep_table; the production lane-sweep instantiations only)
            int nt_mode = 0;
            if (lane_sweep && !vpk && !xwalk && ctx->sweep_table != 0) {
                const unsigned long long stride = ((unsigned long long)ctx->n_lines + 7ull) & ~7ull;
                if (stride * (unsigned long long)ctx->n_shells + 32ull < (1ull << 28)) {
                    if (!ctx->nt_valid) {
                        const long long total = (long long)(stride * (unsigned long long)ctx->n_shells) + 32;  // (+ the slack of a step's loads behind the last row)
                        HIP_TRY(ctx, ctx->nt_t.ensure((size_t)total * 16));
                        HIP_TRY(ctx, ctx->pfx_flag.ensure(sizeof(int)));
                        HIP_TRY(ctx, hipMemsetAsync(ctx->pfx_flag.p, 0, sizeof(int), ctx->stream));
                        hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 65536)), dim3(256), 0, ctx->stream, ctx->nu_line.as<double>(),
                                           ctx->tau_t.as<double>(), ctx->nt_t.as<double2>(), (long long)ctx->n_lines, (long long)ctx->n_shells, (long long)stride, total,
                                           ctx->pfx_flag.as<int>());
                        HIP_TRY(ctx, hipGetLastError());
                        int neg = 0;
                        HIP_TRY(ctx, hipMemcpyAsync(&neg, ctx->pfx_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                        ctx->nt_negative = neg != 0;
                        ctx->nt_stride = (unsigned)stride;
                        ctx->nt_valid = true;
                    }
                }
                if (ctx->nt_valid && !ctx->nt_negative) {
                    nt_mode = (ctx->sweep_table == 2 && !ls3) ? 2 : 1;
                    P.nt_t = ctx->nt_t.as<double>(); P.nt_stride = ctx->nt_stride;
                    if (ls3) kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 3, 1> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 3, 1>;
                    else if (nt_mode == 2) kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 4, 2> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 4, 2>;
                    else kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 4, 1> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 4, 1>;
                }
            }
#undef TMC_PICKLS2
#undef TMC_PICKW3
#undef TMC_PICKLS
#undef TMC_PICKW2
#undef TMC_PICKW
            const long long n = ctx->n_packets;
            ctx->progress_wave = true;  // (one packet supply for the whole call: next_packet counts the packets handed out)
            // (volley queue: more waves than the chip holds at once -- a wave that suspends frees its slot, and the more packets are
            // in flight the more v-packets every tracer launch has to spread over its lanes)
            // CU partition (pass_cus): only for calls long enough to run as several epochs -- the passes of an epoch then have the next one to hide behind
            const int n_xcd = 8;  // gfx950: 8 XCDs x 32 CUs; the bits of a queue's CU mask are interleaved over the XCDs (bit k -> XCD k % 8)
            const bool cu_split = ctx->pass_cus > 0 && !vq && ctx->log_sets != 1 && cus == 32 * n_xcd && n >= 30000000LL;
            const int cus_prop = cu_split ? cus - n_xcd * ctx->pass_cus : cus;
            const int waves = (int)std::max<long long>(1, std::min<long long>((n + 63) / 64, (long long)cus_prop * std::min(wave_waves_per_cu, wide ? 8 : (ls3 ? 12 : 16)) * (vq ? ctx->vq_oversubscribe : 1)));
            // ---- the line-visit log (estimator_log.hpp): two buffer sets, one region per wave; an epoch ends when the regions
            // are full.  Sized for the whole call when that fits log_capacity (1.2x the traces per packet measured in the last
            // call, 128 per packet before anything was measured), else log_capacity.
            if (ctx->events_host && ctx->ev_events && hipEventQuery(ctx->ev_events) == hipSuccess && ctx->events_host[1] > 0)
                ctx->traces_per_packet = (double)ctx->events_host[0] / (double)ctx->events_host[1];
            if (1.1 * ctx->traces_per_packet > ctx->log_budget_per_packet) ctx->log_budget_per_packet = 1.3 * ctx->traces_per_packet;
            const int tiles = std::max((ctx->n_lines + mc::EST_TILE - 1) / mc::EST_TILE, 1);
            const int n_bins = ctx->n_shells * tiles;
            // est_pipeline 1 (estimator_partition.hpp): the records are grouped by shell, then by bin; needs a shell's bins and all shells
            // to fit the partition kernel's local buckets
            const bool partition = ctx->est_pipeline == 1 && tiles <= mc::PART_LOCAL_BUCKETS && ctx->n_shells <= mc::PART_LOCAL_BUCKETS;
            // the shell-sorted log: instantiated for the two production lane-sweep kernels (sixteen waves on the interleaved table, twelve on the separate ones)
            const bool shell_log = lane_sweep && !vpk && !xwalk && !vq && partition && ctx->log_by_shell != 0 && ctx->n_shells <= 64 &&
                                   ((!ls3 && nt_mode == 1) || (ls3 && nt_mode == 0));
            if (shell_log) {
                if (ls3) kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 3, 0, true> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 3, 0, true>;
                else kw = trk ? mc::propagate_wave_kernel<false, true, 16, false, true, false, 4, 1, true> : mc::propagate_wave_kernel<false, false, 16, false, true, false, 4, 1, true>;
                wave_lds = mc::wave_kernel_lds_bytes<false, false, true, true>(ctx->n_shells);
            }
            long long log_capacity = ctx->log_capacity;
            // One log set or two.  Two let the passes of an epoch run on a second stream beside the next launch -- but they do not fit beside sixteen resident waves per CU,
            // so "beside" means: contending with the next launch's first 0.1 s, both slower for it.  Measured at 1e8 packets (profiles/r06_log_sets.txt): the passes before
            // the next launch, alone on the chip, are faster in total, and one set leaves room for epochs half as many again (three launches instead of five): -0.5 %.
            // So a call of many epochs uses one set; shorter calls keep two (the passes of the bulk run beside the drain of the last launch).
            bool one_set = ctx->log_sets == 1;
            if (ctx->log_sets == 0 && partition && !vq && !vpk && ctx->drain_split == 0 && ctx->drain_compact == 0 && ctx->epoch_split == 0 && ctx->pass_cus == 0) {
                double two_set_capacity = (double)ctx->log_capacity;
                size_t free_b = 0, total_b = 0;
                if (!ctx->log_capacity_user && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    const double have = (double)(ctx->log_records[0].cap + ctx->log_records[1].cap + ctx->log_keys[0].cap + ctx->log_keys[1].cap +
                                                 ctx->log_sorted[0].cap + ctx->log_sorted[1].cap + ctx->log_part.cap);
                    two_set_capacity = std::min(two_set_capacity, 0.6 * ((double)free_b + have) / 80.0);
                }
                // (four epochs or more with two sets; at two or three the passes of the first epochs still find room beside the last launch's drain: 4e7 packets
                // 1 292 - 1 308 ms with two sets, 1 320 - 1 351 with one)
                const double per_packet = ctx->traces_per_packet > 0.0 ? 1.05 * ctx->traces_per_packet : ctx->log_budget_per_packet;
                one_set = (double)n * per_packet > 3.0 * two_set_capacity;
            }
            if (one_set && !ctx->log_capacity_user) {
                // (the second set of an earlier, smaller call is given back first; 24 + 4 bytes per record and the 24 of the scratch copy)
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->stream2) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
                ctx->log_records[1].release(); ctx->log_keys[1].release(); ctx->log_sorted[1].release(); ctx->log_bins[1].release(); ctx->log_cursor[1].release();
                size_t free_b = 0, total_b = 0;
                log_capacity = 4000000000LL;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    const double have = (double)(ctx->log_records[0].cap + ctx->log_keys[0].cap + ctx->log_sorted[0].cap + ctx->log_part.cap);
                    log_capacity = std::min<long long>(log_capacity, (long long)(0.6 * ((double)free_b + have) / (partition ? 52.0 : 32.0)));
                }
            } else
            if (!ctx->log_capacity_user) {
                // (fewer, longer epochs are faster -- 25.8 vs 24.5 Mpkt/s at 1e8 packets with 2.5e9 instead of 1.5e9 records per set
                // -- but two sets of 2.5e9 records are 160 GB: never take more than 60 % of what is free, counting what the log holds already)
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    const double have = (double)(ctx->log_records[0].cap + ctx->log_records[1].cap + ctx->log_keys[0].cap + ctx->log_keys[1].cap +
                                                 ctx->log_sorted[0].cap + ctx->log_sorted[1].cap + ctx->log_part.cap);
                    // (bytes per record of capacity: two sets of 24 + 4 and an index of 4 each, or the shared scratch copy of 24)
                    log_capacity = std::min<long long>(log_capacity, (long long)(0.6 * ((double)free_b + have) / (partition ? 80.0 : 64.0)));
                }
            }
            unsigned long long cap = std::min<unsigned long long>((unsigned long long)log_capacity,
                                                                  (unsigned long long)((double)n * ctx->log_budget_per_packet) + 64ull * (unsigned long long)waves + 65536ull);
            if (n_bins > mc::EST_MAX_BINS) cap = 0;  // too many tiles for the LDS histogram: the kernel adds its terms directly
            cap = std::min<unsigned long long>(cap, 0xfffffff0ull);
            // The log is a pool of chunks the waves take one after the other (EstimatorLog, mc_device.hpp): chunks of up to 4096 records
            // (~90 passes of a wave: one pool atomic per 7 ms), at least four per wave on average so that the pool runs dry for all
            // waves at nearly the same time, never fewer than one per wave; a chunk holds >= 256 records (a pass appends up to 64).
            unsigned region_capacity = 0;  // records per chunk
            unsigned long long n_chunks = 0;
            if (cap > 0) {
                // (shell-sorted log: a wave holds an open chunk for every shell -- the pool needs a few more chunks per wave than there are shells; chunks of
                // 2048 records are what the partition kernel stages at a time.  A caller's own log_capacity (tests) is respected: with fewer chunks than that
                // the waves that find the pool empty suspend at once and the call takes more epochs)
                const unsigned long long per_wave = shell_log ? (unsigned long long)(ctx->n_shells + 4) : 4ull;
                region_capacity = ctx->log_chunk_records > 0 ? (unsigned)ctx->log_chunk_records : (shell_log ? 2048u : 4096u);
                while (region_capacity > 256 && (unsigned long long)region_capacity * per_wave * (unsigned long long)waves > cap) region_capacity >>= 1;
                region_capacity &= ~1u;  // even: a chunk of 24-byte records then starts on a 16-byte boundary (partition_kernel stages with 16-byte loads)
                n_chunks = std::max<unsigned long long>(cap / region_capacity, (unsigned long long)waves * ((shell_log && !ctx->log_capacity_user) ? per_wave : 1ull));
                if (n_chunks * region_capacity > 0xfffffff0ull) n_chunks = 0xfffffff0ull / region_capacity;
            }
            // (waves take chunks dynamically: every launch can suspend)
            const bool may_suspend = region_capacity > 0 || vq;
            // a second buffer set (the estimator passes of an epoch overlap the next epoch) only when the call may need several epochs
            // ... or splits off its drain (WaveCold::drain_split): worth a second launch once the call is long enough for a drain to form
            const bool want_split = ctx->drain_split && !vq && !one_set && region_capacity > 0 && n >= 64LL * waves * 4;
            // Tail split: the last epoch of a call should hold only the DRAIN (the ~4 % of the records the longest-lived packets log
            // after the packet supply has run out, on a mostly idle chip), so that the passes over everything before it run beside the
            // drain and only the passes of the tail -- milliseconds -- are left for after the call.  The host knows the call's records
            // from the last call's traces per packet and hands the second-to-last epoch a pool of exactly "what is left minus the
            // tail"; with the chunk pool that epoch ends for all waves at once.  Also for calls whose log fits ONE epoch (their passes
            // were not overlapped with anything before).  A wrong estimate only moves the boundary.
            // (the tail: what the packets in flight when the supply runs out still log -- lanes x ~8 packets' worth of traces, the mean
            // residual life of a heavy-tailed population; only for calls whose passes are worth a second launch: >= 5e8 records)
            const double tail_records = (double)ctx->log_tail_packets * ctx->traces_per_packet * 64.0 * (double)waves;
            // Measured (profiles/r04_tail_split.txt): calls whose log fits one epoch -3 ... -5 % (1e7 - 2e7 packets: their passes ran
            // after the call before); calls of several epochs +0.5 % (their passes overlap the next epoch already, the extra launch
            // costs) -- so only the former.
            const bool one_epoch = (double)region_capacity * (double)n_chunks >= (double)n * ctx->traces_per_packet * 1.05;
            const bool tail_plan = ctx->log_tail_split && !vq && !one_set && region_capacity > 0 && ctx->traces_per_packet > 0.0 && one_epoch &&
                                   (double)n * ctx->traces_per_packet >= 5e8 && (double)n * ctx->traces_per_packet > 2.0 * tail_records;
            // (the second buffer set is allocated as soon as a tail split MAY be planned -- the first call of a context has no estimate yet
            // -- so that no later call of the same size allocates tens of GB in the middle of an iteration)
            const bool tail_possible = ctx->log_tail_split && !vq && !one_set && region_capacity > 0 &&
                                       (double)n * std::max(ctx->traces_per_packet, 16.0) >= 5e8;
            // ... or packs the drain's live lanes into fewer waves (drain_compact): the passes of the launch before run beside the packed drain
            const bool want_compact = ctx->drain_compact > 0 && !vq && !vpk && !cu_split && !shell_log && !one_set && region_capacity > 0 && n > 64LL * (waves - 1) && waves >= 8;
            const int n_sets = (one_set || vq) ? 1 : ((tail_plan || tail_possible || want_split || want_compact || (region_capacity > 0 && (unsigned long long)region_capacity * n_chunks < (unsigned long long)((double)n * ctx->log_budget_per_packet))) ? 2 : 1);
            bool split_armed = want_split && !want_compact;
            bool compact_armed = want_compact;
            int waves_cur = waves;  // (the grid of the next launch: smaller after a compaction)
            mc::LaneSave *cur_save = nullptr; mc::WaveSave *cur_wsave = nullptr; uint32_t *cur_states = nullptr;  // (set below, once the buffers exist)
            ctx->compactions = 0;
            const size_t set_records = (size_t)std::max<unsigned long long>((unsigned long long)region_capacity * n_chunks, 1);
            // (a two-set call on a context whose first set was sized by a larger one-set call: both sets lie in the first set's buffers -- a second allocation of
            // tens of GB costs ~1 s the first time, the memory is cleared)
            const bool set1_inside = n_sets == 2 && !ctx->log_records[1].p && ctx->log_records[0].cap >= 2 * set_records * sizeof(mc::LineVisitRecord) &&
                                     ctx->log_keys[0].cap >= 2 * set_records * sizeof(unsigned) && (partition || ctx->log_sorted[0].cap >= 2 * set_records * sizeof(unsigned));
            auto set_records_ptr = [&](int b) { return set1_inside ? ctx->log_records[0].as<mc::LineVisitRecord>() + (size_t)b * set_records : ctx->log_records[b].as<mc::LineVisitRecord>(); };
            auto set_keys_ptr = [&](int b) { return set1_inside ? ctx->log_keys[0].as<unsigned>() + (size_t)b * set_records : ctx->log_keys[b].as<unsigned>(); };
            auto set_sorted_ptr = [&](int b) { return set1_inside ? ctx->log_sorted[0].as<unsigned>() + (size_t)b * set_records : ctx->log_sorted[b].as<unsigned>(); };
            for (int b = 0; b < n_sets; ++b) {
                if (!(set1_inside && b == 1)) {
                    HIP_TRY(ctx, ctx->log_records[b].ensure(set_records * sizeof(mc::LineVisitRecord)));
                    HIP_TRY(ctx, ctx->log_keys[b].ensure(set_records * sizeof(unsigned)));
                    if (!partition) HIP_TRY(ctx, ctx->log_sorted[b].ensure(set_records * sizeof(unsigned)));
                }
                HIP_TRY(ctx, ctx->log_bins[b].ensure((size_t)(4 * (n_bins + 2) + ctx->n_shells + 2) * sizeof(unsigned)));
                HIP_TRY(ctx, ctx->log_cursor[b].ensure((size_t)(n_chunks + 2) * sizeof(unsigned)));  // chunk counts | pool counter
            }
            if (partition) HIP_TRY(ctx, ctx->log_part.ensure(set_records * sizeof(mc::LineVisitRecord)));
            if (n_sets == 2 && !ctx->stream2) {
                int prio_lo = 0, prio_hi = 0;  // (the estimator passes' stream: highest priority, their workgroups are dispatched first)
                (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
                HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_hi));
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            const bool cu_masked = cu_split && n_sets == 2;
            if (cu_masked && ctx->pass_cus_built != ctx->pass_cus) {
                if (ctx->stream_prop_m) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_prop_m)); HIP_TRY(ctx, hipStreamDestroy(ctx->stream_prop_m)); ctx->stream_prop_m = nullptr; }
                if (ctx->stream_pass_m) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_pass_m)); HIP_TRY(ctx, hipStreamDestroy(ctx->stream_pass_m)); ctx->stream_pass_m = nullptr; }
                uint32_t m_prop[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m_pass[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int bit = 0; bit < cus; ++bit) {  // the first cus_prop bits = (32 - pass_cus) CUs of every XCD
                    if (bit < cus_prop) m_prop[bit >> 5] |= 1u << (bit & 31);
                    else m_pass[bit >> 5] |= 1u << (bit & 31);
                }
                HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&ctx->stream_prop_m, 8, m_prop));
                HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&ctx->stream_pass_m, 8, m_pass));
                if (!ctx->ev_fork_m) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork_m, hipEventDisableTiming));
                if (!ctx->ev_join_m) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join_m, hipEventDisableTiming));
                ctx->pass_cus_built = ctx->pass_cus;
            }
            for (int k = 0; k < 4; ++k)
                if (!ctx->ev_post[k]) HIP_TRY(ctx, hipEventCreate(&ctx->ev_post[k]));
            // ---- per-lane MT19937 state buffers, launch records, suspended lanes
            HIP_TRY(ctx, ctx->seeded_states.ensure((size_t)waves * 64 * mc::WV_STATE_STRIDE * sizeof(uint32_t)));
            HIP_TRY(ctx, ctx->seed_chk[0].ensure((size_t)std::max<long long>(n, 1) * sizeof(mc::LaunchRec)));
            HIP_TRY(ctx, ctx->lane_save.ensure((size_t)waves * 64 * sizeof(mc::LaneSave)));
            HIP_TRY(ctx, ctx->wave_save.ensure((size_t)waves * sizeof(mc::WaveSave)));
            HIP_TRY(ctx, ctx->suspended_dev.ensure(4 * sizeof(unsigned)));
            cur_save = ctx->lane_save.as<mc::LaneSave>(); cur_wsave = ctx->wave_save.as<mc::WaveSave>(); cur_states = ctx->seeded_states.as<uint32_t>();
            if (want_compact) HIP_TRY(ctx, ctx->drain_census.ensure(8 * sizeof(unsigned)));
            if (!ctx->suspended_host) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->suspended_host, 16 * sizeof(unsigned), hipHostMallocDefault));
            if (vq) {
                HIP_TRY(ctx, ctx->vq_req.ensure((size_t)waves * 64 * sizeof(mc::VolleyRequest)));
                HIP_TRY(ctx, ctx->vq_items.ensure((size_t)waves * 64 * mc::VP_ROUND * sizeof(unsigned)));
                HIP_TRY(ctx, ctx->vq_count.ensure(2 * sizeof(unsigned)));
                HIP_TRY(ctx, ctx->vq_jsave.ensure((size_t)waves * 2 * (size_t)ctx->n_shells * sizeof(double)));
                if ((size_t)waves * 64 >= (1u << 29)) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "too many lanes for the volley queue's item words");
            }
            if (vpk) HIP_TRY(ctx, ctx->vp_scratch[0].ensure((size_t)waves * 64 * mc::VP_ROUND * sizeof(mc::VpResult)));
            if (vpk && ctx->vp_carry_min_active > 0) HIP_TRY(ctx, ctx->vp_park.ensure((size_t)waves * 64 * sizeof(mc::VpPark)));
            HIP_TRY(ctx, ctx->wave_cold_dev.ensure(4 * sizeof(mc::WaveCold)));  // (per epoch parity: the launch's block and the second launch's of a split epoch)
            ctx->wave_cold_host.resize(2);
            for (hipEvent_t &e : ctx->ev_split)
                if (!e) HIP_TRY(ctx, hipEventCreate(&e));
            hipStream_t st = ctx->stream;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_start, st));
            if (tune_slot >= 0) HIP_TRY(ctx, hipEventRecord(ctx->ev_tune[0], st));
            if (cu_masked) {  // everything of this call runs on the masked stream from here on; it is joined to the engine's stream at the end
                HIP_TRY(ctx, hipEventRecord(ctx->ev_fork_m, ctx->stream));
                st = ctx->stream_prop_m;
                HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_fork_m, 0));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[0], st));
            if (n > 0) {
                // lazy seeding: only word 397 of every start state is precomputed; the refills continue the init_genrand chains
                mc::LaunchPrepArgs la{};
                la.r0 = F.r0; la.mu0 = F.mu0; la.nu0 = F.nu0; la.e0 = F.e0; la.nu_line = P.nu_line;
                la.seeds = ctx->seeds.as<uint32_t>();
                la.bucket_first = P.bucket_first; la.bucket_shift = P.bucket_shift; la.bucket_n = P.bucket_n; la.n_lines = P.n_lines;
                la.bucket_kmin = P.bucket_kmin; la.t_exp = P.t_exp;
                la.out = ctx->seed_chk[0].as<mc::LaunchRec>(); la.first = 0; la.count = n;
                if (full) hipLaunchKernelGGL(mc::launch_prep_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, la);
                else hipLaunchKernelGGL(mc::launch_prep_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, la);
                HIP_TRY(ctx, hipGetLastError());
            }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[1], st));
            mc::WaveHot hot{};
            hot.nu_line = P.nu_line; hot.tau_t = nt_mode ? P.nt_t : P.tau_t; hot.n_lines = P.n_lines; hot.n_shells = P.n_shells;
            hot.disable_line_scattering = P.disable_line_scattering; hot.debug_flags = P.debug_flags;
            hot.t_exp = P.t_exp;
            // cut-offs of the sweep / walk phases (lanes still busy when the wave moves on): 8 / 8 by default.  Where most blocks are
            // entered through hot sectors the event phase is shorter and later cut-offs pay: 12 / 12 measured -2.5 % on the heavy-tailed
            // configs[2] tables, +2 % on the uniform ones (profiles/r04_cutoffs.txt) -- hence only there, and never against an option
            const bool mostly_hot = ctx->have_hot && 2 * ctx->n_hot_blocks > (long long)ctx->n_levels;
            hot.ls_min_active = (!ctx->ls_min_active_user && mostly_hot) ? 12 : ctx->ls_min_active;
            hot.ls_max_steps = ctx->ls_max_steps;
            hot.walk_min_active = (!ctx->walk_min_active_user && mostly_hot) ? 16 : ctx->walk_min_active;  // (12 until round 6; with the leaner sweep 16: -1.2 %, profiles/r06_cutoffs.txt)
            hot.vq_min_active = ctx->vq_min_active;
            hot.line_block = P.line_interaction_type != 0 ? P.line_block : nullptr;
            // binning + accumulation of one epoch's line-visit log (estimator_log.hpp)
            auto estimator_passes = [&](const mc::EstimatorLog &lg, int b, hipStream_t es) -> hipError_t {
                if (lg.region_capacity == 0) return hipSuccess;
                unsigned *bin_count = ctx->log_bins[b].as<unsigned>(), *bin_start = bin_count + (n_bins + 1),
                         *bin_fill = bin_start + (n_bins + 1), *slice_start = bin_fill + (n_bins + 1);
                unsigned *sorted = set_sorted_ptr(b);
                hipError_t e = hipMemsetAsync(bin_count, 0, (size_t)(n_bins + 1) * sizeof(unsigned), es);
                if (e != hipSuccess) return e;
                const size_t hist_lds = (size_t)n_bins * sizeof(unsigned);
                if (hist_lds > 64 * 1024) {  // more than the default dynamic-LDS limit: BASELINE config 5 has 100 shells x 245 tiles
                    e = hipFuncSetAttribute((const void *)mc::bin_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
                    if (e != hipSuccess) return e;
                    e = hipFuncSetAttribute((const void *)mc::bin_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
                    if (e != hipSuccess) return e;
                }
                const int bin_blocks = cus * 8;
                hipLaunchKernelGGL(mc::bin_count_kernel, dim3(bin_blocks), dim3(256), hist_lds, es, lg.keys, lg.region_count, lg.n_regions,
                                   lg.region_capacity, n_bins, bin_count);
                hipLaunchKernelGGL(mc::bin_scan_kernel, dim3(1), dim3(256), 0, es, bin_count, n_bins, bin_start, bin_fill, slice_start);
                if (partition) {  // estimator_partition.hpp: by shell into the scratch copy, by bin back into the set's own buffer
                    unsigned *shell_fill = slice_start + (n_bins + 2);
                    mc::LineVisitRecord *scratch = ctx->log_part.as<mc::LineVisitRecord>();
                    auto bits_of = [](int n) { int b = 0; while ((1 << b) < n) ++b; return b; };
                    const mc::LineVisitRecord *binned = lg.records;  // what the accumulate kernel reads
                    if (!shell_log && ctx->est_one_level != 0 && n_bins <= mc::PART_LOCAL_BUCKETS) {
                        // few bins (short line lists: 300 on the tardis_example tables): ONE partition pass, log chunks -> scratch copy by bin.  (partition_kernel<2> with
                        // "one shell of n_bins tiles": every chunk's first bucket is bin 0, a staged segment ranks over all bins)
                        hipLaunchKernelGGL(mc::partition_kernel<2>, dim3(cus * 2), dim3(mc::PART_THREADS), 0, es, lg.records, lg.keys, lg.region_count, lg.n_regions,
                                           lg.region_capacity, (const unsigned *)nullptr, n_bins, ctx->n_lines, bits_of(n_bins), bin_fill, scratch);
                        binned = scratch;
                    } else
                    if (shell_log) {  // the chunks hold one shell each: straight to the partition by bin, into the scratch copy
                        hipLaunchKernelGGL(mc::partition_kernel<2>, dim3(cus * 2), dim3(mc::PART_THREADS), 0, es, lg.records, lg.keys, lg.region_count, lg.n_regions,
                                           lg.region_capacity, (const unsigned *)nullptr, lg.tiles_per_shell, ctx->n_lines, bits_of(mc::PART_LOCAL_BUCKETS), bin_fill, scratch);
                        binned = scratch;
                    } else {
                    hipLaunchKernelGGL(mc::partition_shell_fill_kernel, dim3(1), dim3(256), 0, es, bin_start, lg.tiles_per_shell, ctx->n_shells, shell_fill);
                    const int part_blocks = cus * 2;
                    hipLaunchKernelGGL(mc::partition_kernel<1>, dim3(part_blocks), dim3(mc::PART_THREADS), 0, es, lg.records, lg.keys, lg.region_count,
                                       lg.n_regions, lg.region_capacity, (const unsigned *)nullptr, lg.tiles_per_shell, ctx->n_lines, bits_of(ctx->n_shells),
                                       shell_fill, scratch);
                    hipLaunchKernelGGL(mc::partition_kernel<0>, dim3(part_blocks), dim3(mc::PART_THREADS), 0, es, scratch, (const unsigned *)nullptr,
                                       (const unsigned *)nullptr, 0, 0u, bin_start + n_bins, lg.tiles_per_shell, ctx->n_lines, bits_of(mc::PART_LOCAL_BUCKETS),
                                       bin_fill, lg.records);
                    }
                    if (ctx->est_accumulate == 3) {  // the dyadic hierarchy, a lane per record (accumulate_dyadic_kernel<.., LOOP>: 73 KB of LDS, two workgroups per CU)
                        if (full)
                            hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<true, true, true>), dim3(cus * 2), dim3(64 * mc::ACCD_WAVES), 0, es, binned,
                                               (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                        else
                            hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<false, true, true>), dim3(cus * 2), dim3(64 * mc::ACCD_WAVES), 0, es, binned,
                                               (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                        return hipGetLastError();
                    }
                    if (ctx->est_accumulate == 2) {  // the dyadic hierarchy of block sums (accumulate_dyadic_kernel): one workgroup per CU
                        if (full)
                            hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<true, true>), dim3(cus), dim3(64 * mc::ACCD_WAVES), 0, es, binned,
                                               (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                        else
                            hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<false, true>), dim3(cus), dim3(64 * mc::ACCD_WAVES), 0, es, binned,
                                               (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                        return hipGetLastError();
                    }
                    if (full)
                        hipLaunchKernelGGL((mc::accumulate_blocks_kernel<true, true>), dim3(cus * 2), dim3(64 * mc::ACCB_WAVES), 0, es, binned,
                                           (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    else
                        hipLaunchKernelGGL((mc::accumulate_blocks_kernel<false, true>), dim3(cus * 2), dim3(64 * mc::ACCB_WAVES), 0, es, binned,
                                           (const unsigned *)nullptr, bin_start, slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    return hipGetLastError();
                }
                hipLaunchKernelGGL(mc::bin_scatter_kernel, dim3(bin_blocks), dim3(256), hist_lds, es, lg.keys, lg.region_count, lg.n_regions,
                                   lg.region_capacity, n_bins, bin_fill, sorted);
                if (ctx->est_accumulate == 3) {
                    if (full)
                        hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<true, false, true>), dim3(cus * 2), dim3(64 * mc::ACCD_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    else
                        hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<false, false, true>), dim3(cus * 2), dim3(64 * mc::ACCD_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    return hipGetLastError();
                }
                if (ctx->est_accumulate == 2) {
                    if (full)
                        hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<true, false>), dim3(cus), dim3(64 * mc::ACCD_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    else
                        hipLaunchKernelGGL((mc::accumulate_dyadic_kernel<false, false>), dim3(cus), dim3(64 * mc::ACCD_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    return hipGetLastError();
                }
                if (ctx->est_accumulate == 1) {  // one add per aligned block of 8 lines (accumulate_blocks_kernel)
                    const unsigned acc_blocks = (unsigned)(cus * 2);
                    if (full)
                        hipLaunchKernelGGL((mc::accumulate_blocks_kernel<true, false>), dim3(acc_blocks), dim3(64 * mc::ACCB_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    else
                        hipLaunchKernelGGL((mc::accumulate_blocks_kernel<false, false>), dim3(acc_blocks), dim3(64 * mc::ACCB_WAVES), 0, es, lg.records, sorted, bin_start,
                                           slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                    return hipGetLastError();
                }
                const unsigned acc_blocks = (unsigned)(cus * 3);
                if (full)
                    hipLaunchKernelGGL(mc::accumulate_kernel<true>, dim3(acc_blocks), dim3(64 * mc::ACC_WAVES), 0, es, lg.records, sorted, bin_start,
                                       slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                else
                    hipLaunchKernelGGL(mc::accumulate_kernel<false>, dim3(acc_blocks), dim3(64 * mc::ACC_WAVES), 0, es, lg.records, sorted, bin_start,
                                       slice_start, n_bins, lg.tiles_per_shell, ctx->n_lines, P.nu_line, P.jblue_t, P.edot_t);
                return hipGetLastError();
            };
            ctx->post_pending[0] = ctx->post_pending[1] = false;
            ctx->prop_pending = false;
            // result streaming: only the plain epoch loop (no volley queue, no CU partition), with the tracker unpacking it needs done per range
            const bool streaming = rs_armed && !vq && !cu_masked && may_suspend && n >= ctx->rs_min_packets;
            long long rs_pending_lo = 0, rs_pending_hi = 0;  // a range whose unpacking is queued and whose copy the host still has to issue
            if (streaming) {
                ctx->rs.late_capacity = (unsigned)std::min<long long>(n, (long long)waves * 64 * 16);
                HIP_TRY(ctx, ctx->rs_late.ensure((size_t)ctx->rs.late_capacity * sizeof(unsigned)));
                HIP_TRY(ctx, ctx->rs_late_count.ensure(sizeof(unsigned)));
                HIP_TRY(ctx, hipMemsetAsync(ctx->rs_late_count.p, 0, sizeof(unsigned), st));
                if (!ctx->rs_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->rs_stream, hipStreamNonBlocking));
                if (!ctx->rs_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->rs_ev, hipEventDisableTiming));
                if (!ctx->rs_next_host) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->rs_next_host, sizeof(unsigned long long), hipHostMallocDefault));
            }
            auto rs_copy_pending = [&]() -> hipError_t {  // the host's part of a streamed range: sixteen copies into the caller's arrays, beside the running launch
                if (rs_pending_hi <= rs_pending_lo) return hipSuccess;
                hipError_t e = hipStreamWaitEvent(ctx->rs_stream, ctx->rs_ev, 0);
                void *dev[16];
                per_packet_device_arrays(ctx, dev);
                const size_t off = (size_t)rs_pending_lo * 8, bytes = (size_t)(rs_pending_hi - rs_pending_lo) * 8;
                for (int a = 0; a < 16 && e == hipSuccess; ++a)
                    if (ctx->rs.dst[a] && dev[a]) e = hipMemcpyAsync((char *)ctx->rs.dst[a] + off, (char *)dev[a] + off, bytes, hipMemcpyDeviceToHost, ctx->rs_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->rs_stream);
                ctx->rs.upto = rs_pending_hi;
                rs_pending_lo = rs_pending_hi = 0;
                return e;
            };
            const int max_epochs = 1 << 20;
            // volley queue: on for the bulk of a call; once a launch requests fewer v-packets than keep the tracer's lanes busy the
            // launches are bound by their longest v-packet, not by work -- the rest of the call (the drain of the longest-lived
            // packets) runs in ONE launch with the wave kernel's own pooled volleys
            bool vq_on = vq;
            const long long vq_min_items = ctx->vq_min_items >= 0 ? ctx->vq_min_items : (long long)cus * 4 * 64 * 8;
            bool call_complete = n <= 0;
            int log_gen = 0;  // (volley queue: the chunk pool of the shared log is reset after every run of the estimator passes)
            const double records_est = (double)n * ctx->traces_per_packet;  // (tail split: what the call will log, by the last call's measure)
            double records_done = 0.0;
            for (int epoch = 0; n > 0 && epoch < max_epochs; ++epoch) {
                const int b = n_sets == 2 ? (epoch & 1) : 0;
                hipStream_t es = n_sets == 2 ? ctx->stream2 : st;  // the estimator passes of an epoch run beside the next epoch
                if (ctx->post_pending[b]) {  // this buffer set was used two epochs ago: its estimator passes must be over
                    float ms = 0.f;
                    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_post[2 * b + 1]));
                    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_post[2 * b], ctx->ev_post[2 * b + 1]));
                    ctx->sum_post_ms += ms;
                    ctx->post_pending[b] = false;
                }
                mc::EstimatorLog lg{};
                lg.tiles_per_shell = tiles;
                lg.records = set_records_ptr(b);
                lg.keys = set_keys_ptr(b);
                unsigned long long pool_chunks = n_chunks;
                if (tail_plan) {
                    const double bulk = records_est - records_done - tail_records;  // what is left before the tail
                    if (bulk > 0.0 && bulk <= (double)n_chunks * (double)region_capacity)
                        pool_chunks = std::min<unsigned long long>(n_chunks, std::max<unsigned long long>((unsigned long long)waves * (shell_log ? (unsigned long long)(ctx->n_shells + 4) : 1ull),
                                                                                                           (unsigned long long)(bulk / (double)region_capacity) + 1ull));
                }
                lg.n_regions = (int)pool_chunks;
                lg.region_capacity = region_capacity;
                lg.region_count = ctx->log_cursor[b].as<unsigned>();
                lg.pool_next = lg.region_count + n_chunks;
                // (volley queue: the launches of a call go on appending to the same log regions until one of them is full)
                if (!vq || epoch == 0) HIP_TRY(ctx, hipMemsetAsync(lg.region_count, 0, (size_t)(n_chunks + 1) * sizeof(unsigned), st));
                HIP_TRY(ctx, hipMemsetAsync(ctx->suspended_dev.p, 0, 4 * sizeof(unsigned), st));
                if (vq) HIP_TRY(ctx, hipMemsetAsync(ctx->vq_count.p, 0, 2 * sizeof(unsigned), st));
                mc::WaveCold &wc = ctx->wave_cold_host[epoch & 1];
                wc.P = P; wc.D = F; wc.log = lg; wc.seeded_states = cur_states;
                wc.chunk_first = 0; wc.chunk_count = n;
                wc.launch = ctx->seed_chk[0].as<mc::LaunchRec>();
                wc.vp_scratch = ctx->vp_scratch[0].as<mc::VpResult>();
                wc.vp_park = (vpk && !vq_on && ctx->vp_carry_min_active > 0) ? ctx->vp_park.as<mc::VpPark>() : nullptr;
                wc.vp_carry_min_active = ctx->vp_carry_min_active; wc.vp_pad = 0;
                wc.save = may_suspend ? cur_save : nullptr;
                wc.wsave = may_suspend ? cur_wsave : nullptr;
                wc.resume = epoch > 0 ? 1 : 0;
                wc.drain_split = split_armed ? 64 : (compact_armed ? ctx->drain_compact : 0);  // (the most live lanes a wave whose supply has run out suspends with)
                wc.suspended = ctx->suspended_dev.as<unsigned>();
                wc.vq_req = vq_on ? ctx->vq_req.as<mc::VolleyRequest>() : nullptr;
                wc.vq_items = vq_on ? ctx->vq_items.as<unsigned>() : nullptr;
                wc.vq_count = vq ? ctx->vq_count.as<unsigned>() : nullptr;
                wc.vq_jsave = vq ? ctx->vq_jsave.as<double>() : nullptr;
                wc.log_continue = vq ? 1 : 0;
                wc.log_gen = log_gen;
                mc::WaveCold *wc_dev = ctx->wave_cold_dev.as<mc::WaveCold>() + 2 * (epoch & 1);
                HIP_TRY(ctx, store_value(st, wc_dev, wc));
                // split launch (see epoch_split): the estimator passes of the previous epoch are queued (or running) on the second stream
                // (not the drain launch of a tail-split call: its lanes are the call's critical path, half of them would start behind the bulk's passes)
                const bool split = ctx->epoch_split && !want_compact && !vq && !cu_masked && !tail_plan && n_sets == 2 && epoch > 0 && es != st && ctx->post_pending[b ^ 1] && waves >= 8 * cus;
                const int waves1 = split ? waves / 2 : waves_cur;
                if (split) HIP_TRY(ctx, hipEventRecord(ctx->ev_split[0], st));  // (pool, counters and argument block of this epoch are in place)
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[2], st));
                hipLaunchKernelGGL(kw, dim3(waves1), dim3(64), wave_lds, st, hot, (const mc::WaveCold *)wc_dev);
                HIP_TRY(ctx, hipGetLastError());
                double split_w2 = 0.0;  // share of the grid in the second launch
                if (split) {
                    mc::WaveCold wc2 = wc;  // the same epoch for waves [waves1, waves): every per-wave array starts waves1 waves further on
                    wc2.seeded_states = wc.seeded_states + (size_t)waves1 * 64 * mc::WV_STATE_STRIDE;
                    if (wc.save) wc2.save = wc.save + (size_t)waves1 * 64;
                    if (wc.wsave) wc2.wsave = wc.wsave + waves1;
                    if (wc.vp_scratch) wc2.vp_scratch = wc.vp_scratch + (size_t)waves1 * 64 * mc::VP_ROUND;
                    if (wc.vp_park) wc2.vp_park = wc.vp_park + (size_t)waves1 * 64;
                    HIP_TRY(ctx, hipStreamWaitEvent(es, ctx->ev_split[0], 0));
                    HIP_TRY(ctx, store_value(es, wc_dev + 1, wc2));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_split[1], es));
                    hipLaunchKernelGGL(kw, dim3(waves - waves1), dim3(64), wave_lds, es, hot, (const mc::WaveCold *)(wc_dev + 1));
                    HIP_TRY(ctx, hipGetLastError());
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_split[2], es));
                    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_split[2], 0));  // (what follows on the engine's stream -- the read-back, the next epoch -- follows both)
                    split_w2 = (double)(waves - waves1) / (double)waves;
                }
                if (vq_on) {  // the v-packets this launch requested (the item count is read on the device: an empty list costs a launch)
                    const size_t geo_lds = (size_t)4 * (size_t)ctx->n_shells * sizeof(double);
                    const int tracer_waves = cus * 4 * ctx->vq_tracer_waves_per_simd;
                    if (full) hipLaunchKernelGGL(mc::vpacket_trace_kernel<true>, dim3(tracer_waves), dim3(64), geo_lds, st, (const mc::WaveCold *)wc_dev);
                    else hipLaunchKernelGGL(mc::vpacket_trace_kernel<false>, dim3(tracer_waves), dim3(64), geo_lds, st, (const mc::WaveCold *)wc_dev);
                    HIP_TRY(ctx, hipGetLastError());
                }
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[3], st));
                if (may_suspend) {  // (read back before the estimator passes are queued: the host learns early whether another epoch follows)
                    HIP_TRY(ctx, hipMemcpyAsync(ctx->suspended_host, ctx->suspended_dev.p, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    if (vq) HIP_TRY(ctx, hipMemcpyAsync(ctx->suspended_host + 4, ctx->vq_count.p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    if (region_capacity > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->suspended_host + 6, lg.pool_next, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    if (streaming) HIP_TRY(ctx, hipMemcpyAsync(ctx->rs_next_host, ctx->next_packet.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[4], st));
                }
                ctx->launches += 1;
                if (vq) {
                    // volley queue: the estimator passes run when a wave reports a full log region, and once at the end of the call
                    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_chunk[4]));
                    float ms = 0.f;
                    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_chunk[2], ctx->ev_chunk[3]));
                    ctx->sum_prop_ms += ms;
                    const bool last = ctx->suspended_host[0] == 0;
                    if (last || ctx->suspended_host[1] > 0) {
                        HIP_TRY(ctx, hipEventRecord(ctx->ev_post[0], st));
                        HIP_TRY(ctx, estimator_passes(lg, 0, st));
                        HIP_TRY(ctx, hipEventRecord(ctx->ev_post[1], st));
                        HIP_TRY(ctx, hipMemsetAsync(lg.region_count, 0, (size_t)(n_chunks + 1) * sizeof(unsigned), st));
                        ++log_gen;
                        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_post[1]));
                        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_post[0], ctx->ev_post[1]));
                        ctx->sum_post_ms += ms;
                    }
                    if (last) { call_complete = true; break; }
                    if (vq_on && (long long)ctx->suspended_host[4] < vq_min_items) vq_on = false;
                    continue;
                }
                if (cu_masked) {
                    // CU partition: the host first learns whether this was the last epoch.  If not, its passes run on the pass stream's CUs beside
                    // the next epoch; the last epoch's passes take the propagation stream (its CUs are idle now) -- after the passes still
                    // running on the pass stream, with which they share the scratch copy of the records
                    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_chunk[4]));
                    float pms = 0.f;
                    HIP_TRY(ctx, hipEventElapsedTime(&pms, ctx->ev_chunk[2], ctx->ev_chunk[3]));
                    ctx->sum_prop_ms += pms;
                    const bool last = *ctx->suspended_host == 0;
                    hipStream_t ps = ctx->stream_pass_m;
                    if (last) {
                        HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream_pass_m));
                        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
                        ps = st;
                    } else HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->ev_chunk[3], 0));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_post[2 * b], ps));
                    HIP_TRY(ctx, estimator_passes(lg, b, ps));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_post[2 * b + 1], ps));
                    ctx->post_pending[b] = true;
                    if (last) { call_complete = true; break; }
                    records_done += (double)std::min<unsigned long long>((unsigned long long)ctx->suspended_host[6], pool_chunks) * (double)region_capacity;
                    if (ctx->suspended_host[2] > 0) split_armed = false;
                    continue;
                }
                if (es != st) HIP_TRY(ctx, hipStreamWaitEvent(es, ctx->ev_chunk[3], 0));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_post[2 * b], es));
                HIP_TRY(ctx, estimator_passes(lg, b, es));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_post[2 * b + 1], es));
                ctx->post_pending[b] = true;
                ctx->prop_pending = true;
                if (!may_suspend) { call_complete = true; break; }  // (no log: the kernel adds its terms directly and never suspends; the call stays asynchronous)
                // (result streaming: the range unpacked before this launch is copied to the caller's arrays now, while the launch runs)
                if (streaming) HIP_TRY(ctx, rs_copy_pending());
                // is anything suspended?  (the only host synchronisation of a call: once per epoch)
                HIP_TRY(ctx, hipEventSynchronize(ctx->ev_chunk[4]));
                float ms = 0.f;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_chunk[2], ctx->ev_chunk[3]));
                if (split_w2 > 0.0) {  // a split epoch counts with the wave-weighted duration of its two launches (= the time a launch of the whole grid stands for)
                    float ms2 = 0.f;
                    HIP_TRY(ctx, hipEventElapsedTime(&ms2, ctx->ev_split[1], ctx->ev_split[2]));
                    ms = (float)((1.0 - split_w2) * (double)ms + split_w2 * (double)ms2);
                }
                ctx->sum_prop_ms += ms;
                ctx->prop_pending = false;
                if (*ctx->suspended_host == 0) { call_complete = true; break; }
                records_done += (double)std::min<unsigned long long>((unsigned long long)ctx->suspended_host[6], pool_chunks) * (double)region_capacity;
                if (ctx->suspended_host[2] > 0) split_armed = false;  // (the drain has been split off: the next launch runs to the end)
                if (compact_armed && ctx->suspended_host[2] > 0) {
                    // some waves have suspended with few live lanes: what is left on the grid?
                    unsigned *cen = ctx->drain_census.as<unsigned>();
                    HIP_TRY(ctx, hipMemsetAsync(cen, 0, 8 * sizeof(unsigned), st));
                    hipLaunchKernelGGL(mc::drain_census_kernel, dim3(waves_cur), dim3(64), 0, st, cur_save, cur_wsave, waves_cur, n, cen);
                    HIP_TRY(ctx, hipGetLastError());
                    HIP_TRY(ctx, hipMemcpyAsync(ctx->suspended_host + 8, cen, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    HIP_TRY(ctx, hipStreamSynchronize(st));
                    const unsigned live = ctx->suspended_host[8], reserved = ctx->suspended_host[9], waiting = ctx->suspended_host[11];
                    const unsigned density = (unsigned)ctx->drain_pack_lanes;
                    const int packed = (int)((live + density - 1u) / density);
                    // (only when nothing is left to hand out, and when it frees more than half of the grid)
                    if (reserved == 0 && waiting == 0 && live > 0 && 2 * packed <= waves_cur) {
                        const int g = ctx->compactions & 1;
                        HIP_TRY(ctx, ctx->lane_save_c[g].ensure((size_t)packed * 64 * sizeof(mc::LaneSave)));
                        HIP_TRY(ctx, ctx->wave_save_c[g].ensure((size_t)packed * sizeof(mc::WaveSave)));
                        HIP_TRY(ctx, ctx->seeded_states_c[g].ensure((size_t)packed * 64 * mc::WV_STATE_STRIDE * sizeof(uint32_t)));
                        HIP_TRY(ctx, hipMemsetAsync(cen + 4, 0, sizeof(unsigned), st));
                        hipLaunchKernelGGL(mc::drain_compact_kernel, dim3(waves_cur), dim3(64), 0, st, (const mc::LaneSave *)cur_save, (const mc::WaveSave *)cur_wsave,
                                           (const uint32_t *)cur_states, waves_cur, ctx->lane_save_c[g].as<mc::LaneSave>(), ctx->seeded_states_c[g].as<uint32_t>(), cen + 4, density);
                        HIP_TRY(ctx, hipGetLastError());
                        hipLaunchKernelGGL(mc::drain_compact_finish_kernel, dim3((unsigned)((packed * 64 + 255) / 256)), dim3(256), 0, st, ctx->lane_save_c[g].as<mc::LaneSave>(),
                                           ctx->wave_save_c[g].as<mc::WaveSave>(), (const unsigned *)(cen + 4), n, density);
                        HIP_TRY(ctx, hipGetLastError());
                        cur_save = ctx->lane_save_c[g].as<mc::LaneSave>(); cur_wsave = ctx->wave_save_c[g].as<mc::WaveSave>(); cur_states = ctx->seeded_states_c[g].as<uint32_t>();
                        waves_cur = packed;
                        ctx->compactions += 1;
                        if (packed < 2 * cus || (int)density <= 2 * ctx->drain_compact) compact_armed = false;  // (nothing left worth freeing / the packed waves would suspend again at once)
                    }
                }
                if (streaming) {
                    // packets [0, handed) have been handed out; those of them still in flight (suspended lanes, reserved blocks) go on the late list, the range
                    // [upto, handed) is unpacked now -- in front of the next launch on the same stream -- and copied by the host beside that launch
                    const long long handed = (long long)std::min<unsigned long long>(*ctx->rs_next_host, (unsigned long long)n);
                    if (handed - ctx->rs.upto >= ctx->rs_min_packets) {
                        hipLaunchKernelGGL(mc::late_list_kernel, dim3(waves_cur), dim3(64), 0, st, (const mc::LaneSave *)cur_save, (const mc::WaveSave *)cur_wsave, waves_cur, ctx->rs.upto, handed,
                                           ctx->rs_late.as<unsigned>(), ctx->rs_late_count.as<unsigned>(), ctx->rs.late_capacity);
                        HIP_TRY(ctx, hipGetLastError());
                        if (ctx->track) {
                            const long long cnt = handed - ctx->rs.upto;
                            hipLaunchKernelGGL(mc::tracker_unpack_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, F, ctx->rs.upto, cnt, (const unsigned *)nullptr);
                            HIP_TRY(ctx, hipGetLastError());
                        }
                        HIP_TRY(ctx, hipEventRecord(ctx->rs_ev, st));
                        rs_pending_lo = ctx->rs.upto; rs_pending_hi = handed;
                    }
                }
            }
            if (streaming && call_complete) HIP_TRY(ctx, rs_copy_pending());  // (a range queued before the last launch)
            if (!call_complete)  // (packets would be left suspended in lane_save, outputs and estimators silently incomplete)
                return fail(ctx, TARDIS_MC_ERR_STATE, "propagate: %d launches did not finish the call (waves still suspended)", max_epochs);
            if (cu_masked) {  // both masked streams join the engine's stream
                HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream_pass_m));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_join_m, st));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join_m, 0));
            } else if (n_sets == 2 && (ctx->post_pending[0] || ctx->post_pending[1])) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            }
            ctx->wave_epoch_mode = true;
            ctx->progress_done = true;  // (every packet has been handed out and has ended: the host read "nothing suspended")
            long long unpack_from = 0;
            if (streaming && ctx->rs.upto > 0) {
                // what was streamed: [0, upto) minus the late list.  The list's length is read back here (the call has synchronised with every launch already)
                unsigned n_late = 0;
                HIP_TRY(ctx, hipMemcpyAsync(&n_late, ctx->rs_late_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (n_late <= ctx->rs.late_capacity) {
                    ctx->rs.valid = true;
                    ctx->rs.n_late = n_late;
                    unpack_from = ctx->rs.upto;
                    if (n_late > 0) {
                        if (ctx->track) {
                            hipLaunchKernelGGL(mc::tracker_unpack_kernel, dim3((n_late + 255) / 256), dim3(256), 0, ctx->stream, F, 0LL, (long long)n_late, ctx->rs_late.as<unsigned>());
                            HIP_TRY(ctx, hipGetLastError());
                        }
                        HIP_TRY(ctx, ctx->rs_vals.ensure((size_t)16 * n_late * 8));
                        void *dev[16];
                        per_packet_device_arrays(ctx, dev);
                        for (int a = 0; a < 16; ++a)
                            if (ctx->rs.dst[a] && dev[a]) {
                                hipLaunchKernelGGL(mc::gather64_kernel, dim3((n_late + 255) / 256), dim3(256), 0, ctx->stream, (const unsigned long long *)dev[a],
                                                   ctx->rs_late.as<unsigned>(), (long long)n_late, ctx->rs_vals.as<unsigned long long>() + (size_t)a * n_late);
                                HIP_TRY(ctx, hipGetLastError());
                            }
                    }
                }  // (else: the list overflowed -- get_results copies everything)
            }
            if (ctx->track && n > unpack_from) {  // the wave kernel's tracker records -> the boundary's arrays
                hipLaunchKernelGGL(mc::tracker_unpack_kernel, dim3((unsigned)((n - unpack_from + 255) / 256)), dim3(256), 0, ctx->stream, F, unpack_from, n - unpack_from,
                                   (const unsigned *)nullptr);
                HIP_TRY(ctx, hipGetLastError());
            }
        } else {
            ctx->wave_epoch_mode = false;
            long long chunk = std::min<long long>(std::max<long long>(ctx->n_packets, 1), ctx->chunk_packets);
            HIP_TRY(ctx, ctx->seeded_states.ensure((size_t)chunk * mc::WV_STATE_STRIDE * sizeof(uint32_t)));
            // group size: 8 lanes per packet pays off when the sweeps between events are short (sparse line lists)
            const int G = ctx->group_size == 8 ? 8 : (ctx->group_size == 16 ? 16 : ((ctx->n_lines <= 100000 && !vpk) ? 8 : 16));
            const int block = 256;
            const size_t lds = G == 8 ? mc::group_kernel_lds_bytes<8, 256>(ctx->n_shells) : mc::group_kernel_lds_bytes<16, 256>(ctx->n_shells);
            if (lds > 160 * 1024) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "n_shells too large for the LDS J/nu_bar accumulator");
            const int blocks_per_cu = std::max(1, std::min(std::min(ctx->blocks_per_cu, 8), (int)((160 * 1024) / lds)));
            using KernelFn = void (*)(mc::GroupArgs, uint32_t *, long long, long long);
            KernelFn k;
#define TMC_PICK2(G_, V_) (full ? (trk ? mc::propagate_group_kernel<true, true, G_, 256, 4, V_> : mc::propagate_group_kernel<true, false, G_, 256, 4, V_>) \
                                : (trk ? mc::propagate_group_kernel<false, true, G_, 256, 4, V_> : mc::propagate_group_kernel<false, false, G_, 256, 4, V_>))
            if (G == 16) k = vpk ? TMC_PICK2(16, true) : TMC_PICK2(16, false);
            else k = vpk ? TMC_PICK2(8, true) : TMC_PICK2(8, false);
#undef TMC_PICK2
            hipStream_t st = ctx->stream;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_start, st));
            uint32_t *seeded = ctx->seeded_states.as<uint32_t>();
            for (long long first = 0; first < ctx->n_packets; first += chunk) {
                const long long count = std::min(chunk, ctx->n_packets - first);
                const int ci = ctx->chunks_timed;
                while ((int)ctx->ev_chunk.size() < 4 * (ci + 1)) {
                    hipEvent_t e;
                    HIP_TRY(ctx, hipEventCreate(&e));
                    ctx->ev_chunk.push_back(e);
                }
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[4 * ci], st));
                hipLaunchKernelGGL(mc::seed_states_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                                   ctx->seeds.as<uint32_t>(), seeded, first, count, mc::MT_N);
                HIP_TRY(ctx, hipGetLastError());
                HIP_TRY(ctx, hipMemsetAsync(ctx->next_packet.p, 0, sizeof(unsigned long long), st));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[4 * ci + 1], st));
                const int groups_per_block = block / G;
                long long want_blocks = (count + groups_per_block - 1) / groups_per_block;
                int blocks = (int)std::max<long long>(1, std::min<long long>(want_blocks, (long long)cus * blocks_per_cu));
                // (the group kernel updates the line estimators with atomics: logging its traces was measured and is a loss
                // there -- the record bookkeeping costs its redundant-lane event loop more than the deferred atomics do)
                hipLaunchKernelGGL(k, dim3(blocks), dim3(block), lds, st, P, seeded, first, count);
                HIP_TRY(ctx, hipGetLastError());
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[4 * ci + 2], st));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[4 * ci + 3], st));
                ctx->chunks_timed = ci + 1;
            }
        }
    }
    {   // events per packet of this call, for the log sizing of the next one (asynchronous, pinned host memory)
        if (!ctx->events_host) {
            HIP_TRY(ctx, hipHostMalloc((void **)&ctx->events_host, 2 * sizeof(unsigned long long), hipHostMallocDefault));
            ctx->events_host[0] = ctx->events_host[1] = 0;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_events, hipEventDisableTiming));
        }
        ctx->events_host[1] = (unsigned long long)ctx->n_packets;
        HIP_TRY(ctx, hipMemcpyAsync(&ctx->events_host[0], ctx->counters.as<unsigned long long>() + TARDIS_MC_CNT_EVENTS,
                                    sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_events, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = true;
    if (tune_slot >= 0) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_tune[1], ctx->stream));
        ctx->ls_tune.pending = tune_slot;
    }
    return TARDIS_MC_OK;
}

int tardis_mc_synchronize(TardisMcContext *ctx)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->progress_done = true;
    return TARDIS_MC_OK;
}

int tardis_mc_progress(TardisMcContext *ctx, int64_t *out_packets_started, int64_t *out_packets_total)
{
    if (!ctx || !out_packets_started || !out_packets_total) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    const long long total = ctx->progress_total.load();
    *out_packets_total = total;
    *out_packets_started = 0;
    if (ctx->progress_done.load()) { *out_packets_started = total; return TARDIS_MC_OK; }
    if (!ctx->progress_wave.load()) return TARDIS_MC_OK;
    std::lock_guard<std::mutex> lock(ctx->progress_mutex);
    if (hipSetDevice(ctx->device) != hipSuccess) return TARDIS_MC_ERR_HIP;  // (the polling thread's current device is 0 until it says otherwise)
    if (!ctx->next_packet.p || !ctx->ev_progress_reset || hipEventQuery(ctx->ev_progress_reset) != hipSuccess) return TARDIS_MC_OK;  // (not reset yet)
    if (!ctx->stream_progress && hipStreamCreateWithFlags(&ctx->stream_progress, hipStreamNonBlocking) != hipSuccess) return TARDIS_MC_ERR_HIP;
    if (!ctx->progress_host && hipHostMalloc((void **)&ctx->progress_host, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return TARDIS_MC_ERR_HIP;
    if (hipMemcpyAsync(ctx->progress_host, ctx->next_packet.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream_progress) != hipSuccess ||
        hipStreamSynchronize(ctx->stream_progress) != hipSuccess)
        return TARDIS_MC_ERR_HIP;
    *out_packets_started = (int64_t)std::min<unsigned long long>(*ctx->progress_host, (unsigned long long)std::max<long long>(total, 0));  // (a wave reserves 32 at a time)
    return TARDIS_MC_OK;
}

int tardis_mc_last_propagate_ms(TardisMcContext *ctx, double *out_ms)
{
    if (!ctx || !out_ms) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->timed) return fail(ctx, TARDIS_MC_ERR_STATE, "no propagate has been timed yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    *out_ms = (double)ms;
    return TARDIS_MC_OK;
}

int tardis_mc_last_kernel_times(TardisMcContext *ctx, double *out_seed_ms, double *out_propagate_ms, int *out_launches)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->timed) return fail(ctx, TARDIS_MC_ERR_STATE, "no propagate has been timed yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    double seed = 0.0, prop = 0.0;
    if (ctx->wave_epoch_mode) {  // wave kernel: launch preparation once, then one propagation launch + estimator passes per epoch
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_chunk[0], ctx->ev_chunk[1]));
        ctx->sum_seed_ms = ms;
        if (ctx->prop_pending) {
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_chunk[2], ctx->ev_chunk[3]));
            ctx->sum_prop_ms += ms;
            ctx->prop_pending = false;
        }
        for (int b = 0; b < 2; ++b)
            if (ctx->post_pending[b]) {
                HIP_TRY(ctx, hipEventSynchronize(ctx->ev_post[2 * b + 1]));
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_post[2 * b], ctx->ev_post[2 * b + 1]));
                ctx->sum_post_ms += ms;
                ctx->post_pending[b] = false;
            }
        ctx->last_post_ms = ctx->sum_post_ms;
        if (out_seed_ms) *out_seed_ms = ctx->sum_seed_ms;
        if (out_propagate_ms) *out_propagate_ms = ctx->sum_prop_ms;
        if (out_launches) *out_launches = std::max(ctx->launches, 1);
        return TARDIS_MC_OK;
    }
    if (ctx->chunks_timed == 0) {  // lane-per-packet variant: one launch, no seeding kernel
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
        prop = ms;
    }
    double post = 0.0;
    for (int ci = 0; ci < ctx->chunks_timed; ++ci) {
        float a = 0.f, b = 0.f, c = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev_chunk[4 * ci], ctx->ev_chunk[4 * ci + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev_chunk[4 * ci + 1], ctx->ev_chunk[4 * ci + 2]));
        HIP_TRY(ctx, hipEventElapsedTime(&c, ctx->ev_chunk[4 * ci + 2], ctx->ev_chunk[4 * ci + 3]));
        seed += a;
        prop += b;
        post += c;
    }
    ctx->last_post_ms = post;
    if (out_seed_ms) *out_seed_ms = seed;
    if (out_propagate_ms) *out_propagate_ms = prop;
    if (out_launches) *out_launches = ctx->chunks_timed ? ctx->chunks_timed : 1;
    return TARDIS_MC_OK;
}

int tardis_mc_last_counters(TardisMcContext *ctx, int64_t out_counters[TARDIS_MC_N_COUNTERS])
{
    if (!ctx || !out_counters) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->counters.p) return fail(ctx, TARDIS_MC_ERR_STATE, "no work counters yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long cnt[TARDIS_MC_N_COUNTERS];
    HIP_TRY(ctx, hipMemcpyAsync(cnt, ctx->counters.p, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < TARDIS_MC_N_COUNTERS; ++k) out_counters[k] = (int64_t)cnt[k];
    out_counters[TARDIS_MC_CNT_PACKETS] = ctx->n_packets;
    return TARDIS_MC_OK;
}

int tardis_mc_last_variant(TardisMcContext *ctx) { return ctx ? ctx->last_variant : -1; }
int tardis_mc_last_compactions(TardisMcContext *ctx) { return ctx ? ctx->compactions : -1; }

int tardis_mc_last_estimator_ms(TardisMcContext *ctx, double *out_ms)
{
    if (!ctx || !out_ms) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    double seed, prop;
    int launches;
    int rc = tardis_mc_last_kernel_times(ctx, &seed, &prop, &launches);
    if (rc) return rc;
    *out_ms = ctx->last_post_ms;
    return TARDIS_MC_OK;
}

static int reduce_estimator_copies(TardisMcContext *ctx)
{
    if (ctx->est_copies <= 1) return TARDIS_MC_OK;
    EstLayout e = est_layout(ctx->est_S, ctx->est_L, ctx->est_G, ctx->est_copies);
    long long n = (long long)(2 * ctx->est_S * ctx->est_L);
    hipLaunchKernelGGL(reduce_copies_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->est.as<double>() + e.jblue, n,
                       (long long)e.copy_stride, ctx->est_copies);
    HIP_TRY(ctx, hipGetLastError());
    return TARDIS_MC_OK;
}

int tardis_mc_get_results(TardisMcContext *ctx, TardisMcResult *res)
{
    if (!ctx || !res) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->est_valid) return fail(ctx, TARDIS_MC_ERR_STATE, "nothing to fetch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = reduce_estimator_copies(ctx);
    if (rc) return rc;
    const size_t P = (size_t)ctx->n_packets, S = ctx->est_S, L = ctx->est_L, G = ctx->est_G;
    EstLayout e = est_layout(S, L, G, ctx->est_copies);
    double *base = ctx->est.as<double>();
    auto d2h = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (!dst || !bytes) return hipSuccess;
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    };
    if (ctx->have_packets) {
        // (arrays a streaming propagate has been filling, tardis_mc_stream_results: only what has not been sent yet)
        void *dev[16], *host[16];
        per_packet_device_arrays(ctx, dev);
        per_packet_host_arrays(res, host);
        std::vector<CopyJob> jobs;
        bool patch[16];
        for (int a = 0; a < 16; ++a) {
            patch[a] = false;
            if (!host[a] || !dev[a]) continue;
            if (ctx->rs.valid && host[a] == ctx->rs.dst[a]) {
                patch[a] = ctx->rs.n_late > 0;
                const size_t off = (size_t)ctx->rs.upto * 8;
                if (P * 8 > off) jobs.push_back({(char *)host[a] + off, (char *)dev[a] + off, P * 8 - off});
            } else {
                jobs.push_back({host[a], dev[a], P * 8});
            }
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the propagation and the tracker unpacking are over)
        if (ctx->rs.valid && ctx->rs.n_late > 0) {  // the packets that were in flight when their range was sent
            const size_t m = (size_t)ctx->rs.n_late;
            std::vector<unsigned> idx(m);
            std::vector<unsigned long long> vals(m);
            HIP_TRY(ctx, hipMemcpy(idx.data(), ctx->rs_late.p, m * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int a = 0; a < 16; ++a) {
                if (!patch[a]) continue;
                HIP_TRY(ctx, hipMemcpy(vals.data(), ctx->rs_vals.as<unsigned long long>() + (size_t)a * m, m * 8, hipMemcpyDeviceToHost));
                unsigned long long *out = (unsigned long long *)host[a];
                for (size_t j = 0; j < m; ++j) out[idx[j]] = vals[j];
            }
        }
        HIP_TRY(ctx, host_copy(ctx, jobs, false));
    }
    HIP_TRY(ctx, d2h(res->j_estimator, base + e.J, S * 8));
    HIP_TRY(ctx, d2h(res->nu_bar_estimator, base + e.nubar, S * 8));
    HIP_TRY(ctx, d2h(res->v_packets_energy_hist, base + e.vhist, G * 8));
    if (res->j_blue_estimator || res->edotlu_estimator) {
        HIP_TRY(ctx, ctx->staging.ensure(L * S * sizeof(double)));
        if (res->j_blue_estimator) {
            HIP_TRY(ctx, launch_transpose(ctx->stream, base + e.jblue, ctx->staging.as<double>(), (long long)S, (long long)L));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, host_copy(ctx, {{res->j_blue_estimator, ctx->staging.p, L * S * 8}}, false));
        }
        if (res->edotlu_estimator) {
            HIP_TRY(ctx, launch_transpose(ctx->stream, base + e.edot, ctx->staging.as<double>(), (long long)S, (long long)L));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, host_copy(ctx, {{res->edotlu_estimator, ctx->staging.p, L * S * 8}}, false));
        }
    }
    unsigned long long cnt[TARDIS_MC_N_COUNTERS] = {0};
    if (ctx->counters.p) HIP_TRY(ctx, d2h(cnt, ctx->counters.p, sizeof cnt));
    long long ferr[2] = {0x7fffffffffffffffLL, 0};
    if (ctx->first_error.p) HIP_TRY(ctx, d2h(ferr, ctx->first_error.p, sizeof ferr));
    unsigned long long vcount = 0;
    const bool vlog = ctx->cfg.enable_vpacket_tracking && ctx->cfg.number_of_vpackets > 0 && ctx->vlog_count.p;
    if (vlog) HIP_TRY(ctx, d2h(&vcount, ctx->vlog_count.p, sizeof vcount));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < TARDIS_MC_N_COUNTERS; ++k) res->counters[k] = (int64_t)cnt[k];
    res->counters[TARDIS_MC_CNT_PACKETS] = ctx->n_packets;
    if (ctx->n_packets > 0) ctx->traces_per_packet = (double)cnt[TARDIS_MC_CNT_EVENTS] / (double)ctx->n_packets;
    res->first_error_packet = -1;
    res->error_code = 0;
    if (ferr[0] != 0x7fffffffffffffffLL) {
        res->first_error_packet = ferr[0];
        double marker = 0.0;  // the failing lane stored its error code in out_nu[packet]
        HIP_TRY(ctx, hipMemcpy(&marker, ctx->out_nu.as<double>() + ferr[0], 8, hipMemcpyDeviceToHost));
        res->error_code = (int)marker;
    }
    res->vpacket_log_count = 0;
    if (vlog) {
        res->vpacket_log_count = (int64_t)vcount;
        size_t n = (size_t)std::min<unsigned long long>(vcount, (unsigned long long)ctx->vlog_capacity);
        std::vector<long long> pk(n);
        std::vector<int> seq(n);
        std::vector<double> nu(n), en(n), mu(n), rr(n);
        if (n) {
            HIP_TRY(ctx, hipMemcpy(pk.data(), ctx->vlog_packet.p, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(seq.data(), ctx->vlog_seq.p, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(nu.data(), ctx->vlog_nu.p, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(en.data(), ctx->vlog_energy.p, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(mu.data(), ctx->vlog_mu.p, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(rr.data(), ctx->vlog_r.p, n * 8, hipMemcpyDeviceToHost));
        }
        // the reference consolidates per-packet lists in packet order (packet_collections.py:310-396)
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pk[a] != pk[b] ? pk[a] < pk[b] : seq[a] < seq[b]; });
        size_t m = std::min<size_t>(n, (size_t)std::max<int64_t>(0, res->vpacket_log_capacity));
        for (size_t k = 0; k < m; ++k) {
            size_t s = order[k];
            if (res->vpacket_nus) res->vpacket_nus[k] = nu[s];
            if (res->vpacket_energies) res->vpacket_energies[k] = en[s];
            if (res->vpacket_initial_mus) res->vpacket_initial_mus[k] = mu[s];
            if (res->vpacket_initial_rs) res->vpacket_initial_rs[k] = rr[s];
        }
    }
    return res->error_code;
}

int tardis_mc_stream_results(TardisMcContext *ctx, const TardisMcResult *dst)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    ctx->rs.armed = false;
    if (!dst) return TARDIS_MC_OK;
    per_packet_host_arrays(dst, ctx->rs.dst);
    ctx->rs.armed = true;
    return TARDIS_MC_OK;
}

int tardis_mc_streamed_packets(TardisMcContext *ctx, int64_t *out_streamed, int64_t *out_resent)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (out_streamed) *out_streamed = ctx->rs.valid ? ctx->rs.upto : 0;
    if (out_resent) *out_resent = ctx->rs.valid ? ctx->rs.n_late : 0;
    return TARDIS_MC_OK;
}

int tardis_mc_run(TardisMcContext *ctx, const TardisMcPackets *packets, const TardisMcGeometry *geometry,
                  const TardisMcOpacity *opacity, const TardisMcConfig *config, TardisMcResult *result)
{
    int rc;
    if ((rc = tardis_mc_set_geometry(ctx, geometry))) return rc;
    if ((rc = tardis_mc_set_opacity(ctx, opacity))) return rc;
    if ((rc = tardis_mc_set_config(ctx, config))) return rc;
    // the caller's log capacity is for this call only, whichever way the call ends (the context is cached per process)
    struct VlogScope {
        TardisMcContext *c; bool was_user; long long was;
        ~VlogScope() { c->vlog_capacity_user = was_user; if (was_user) c->vlog_capacity = was; }
    } scope{ctx, ctx->vlog_capacity_user, ctx->vlog_capacity};
    if (result && result->vpacket_log_capacity > 0) { ctx->vlog_capacity = result->vpacket_log_capacity; ctx->vlog_capacity_user = true; }
    if ((rc = tardis_mc_set_packets(ctx, packets))) return rc;
    if ((rc = tardis_mc_reset_estimators(ctx))) return rc;
    if (result && (rc = tardis_mc_stream_results(ctx, result))) return rc;  // (the result arrays are known from the start: filled launch by launch)
    if ((rc = tardis_mc_propagate(ctx))) return rc;
    if ((rc = tardis_mc_synchronize(ctx))) return rc;
    return tardis_mc_get_results(ctx, result);
}

int tardis_mc_packet_spectrum(TardisMcContext *ctx, double time_of_simulation, double luminosity_nu_start,
                              double luminosity_nu_end, double *emitted_luminosity_hist, double *reabsorbed_luminosity_hist,
                              double *out_emitted_luminosity, double *out_reabsorbed_luminosity)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->have_packets || !ctx->have_config || ctx->cfg.n_spectrum_grid < 2)
        return fail(ctx, TARDIS_MC_ERR_STATE, "packet spectrum needs propagated packets and a spectrum grid");
    if (!(time_of_simulation > 0)) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "time_of_simulation must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)ctx->cfg.n_spectrum_grid - 1;
    DevBuf work;
    HIP_TRY(ctx, work.ensure((2 * B + 2) * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(work.p, 0, (2 * B + 2) * sizeof(double), ctx->stream));
    double *w = work.as<double>();
    if (ctx->n_packets > 0) {
        const int blocks = (int)std::min<long long>((ctx->n_packets + 255) / 256, 4096);
        hipLaunchKernelGGL(spectrum_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->out_nu.as<double>(), ctx->out_e.as<double>(),
                           ctx->n_packets, ctx->grid.as<double>(), (int)ctx->cfg.n_spectrum_grid, time_of_simulation,
                           luminosity_nu_start, luminosity_nu_end, w, w + B, w + 2 * B);
        HIP_TRY(ctx, hipGetLastError());
    }
    double lum[2] = {0, 0};
    if (emitted_luminosity_hist) HIP_TRY(ctx, hipMemcpyAsync(emitted_luminosity_hist, w, B * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (reabsorbed_luminosity_hist) HIP_TRY(ctx, hipMemcpyAsync(reabsorbed_luminosity_hist, w + B, B * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(lum, w + 2 * B, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out_emitted_luminosity) *out_emitted_luminosity = lum[0];
    if (out_reabsorbed_luminosity) *out_reabsorbed_luminosity = lum[1];
    work.release();
    return TARDIS_MC_OK;
}

int tardis_mc_radiation_field(TardisMcContext *ctx, double time_of_simulation, const double *volume, double w_epsilon,
                              int detailed_optical_window, double *t_radiative, double *dilution_factor, double *j_blues)
{
    if (!ctx || !volume) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->est_valid || !ctx->have_opacity || !ctx->have_geometry)
        return fail(ctx, TARDIS_MC_ERR_STATE, "radiation field update needs propagated estimators");
    if (!(time_of_simulation > 0)) return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "time_of_simulation must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = reduce_estimator_copies(ctx);
    if (rc) return rc;
    const size_t S = ctx->est_S, L = ctx->est_L;
    EstLayout e = est_layout(S, L, ctx->est_G, ctx->est_copies);
    double *base = ctx->est.as<double>();
    // constants, tardis/constants.py:1 (CODATA 2010, cgs)
    const double h = 6.62606957e-27, k_b = 1.3806488e-16, sigma_sb = 5.670373e-5, c = mc::C_LIGHT, zeta5 = 1.0369277551433699;
    const double pi = 3.141592653589793;
    RadFieldConsts k;
    k.t_rad_const = (pi * pi * pi * pi / (15 * 24 * zeta5)) * (h / k_b);
    k.four_sigma = 4 * sigma_sb;
    k.jblue_norm_num = c * ctx->t_exp;
    k.four_pi_tsim = 4 * pi * time_of_simulation;
    k.tsim = time_of_simulation;
    k.planck_coef = 2 * h / (c * c);
    k.h = h; k.k_b = k_b; k.w_epsilon = w_epsilon; k.c_ang = c * 1e8;
    DevBuf work, out_t;
    HIP_TRY(ctx, work.ensure(4 * S * sizeof(double)));
    double *d_vol = work.as<double>(), *d_t = d_vol + S, *d_w = d_t + S, *d_norm = d_w + S;
    HIP_TRY(ctx, hipMemcpyAsync(d_vol, volume, S * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(radfield_shell_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, ctx->stream, base + e.J, base + e.nubar,
                       d_vol, (int)S, k, d_t, d_w, d_norm);
    HIP_TRY(ctx, hipGetLastError());
    if (t_radiative) HIP_TRY(ctx, hipMemcpyAsync(t_radiative, d_t, S * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (dilution_factor) HIP_TRY(ctx, hipMemcpyAsync(dilution_factor, d_w, S * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (j_blues && L > 0) {
        HIP_TRY(ctx, out_t.ensure(L * S * sizeof(double)));
        HIP_TRY(ctx, ctx->staging.ensure(L * S * sizeof(double)));
        const unsigned bx = (unsigned)std::min<size_t>((L + 255) / 256, 1024);
        hipLaunchKernelGGL(radfield_jblue_kernel, dim3(bx, (unsigned)S), dim3(256), 0, ctx->stream, base + e.jblue,
                           ctx->nu_line.as<double>(), d_t, d_w, d_norm, (int)S, (long long)L, k, detailed_optical_window,
                           out_t.as<double>());
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, launch_transpose(ctx->stream, out_t.as<double>(), ctx->staging.as<double>(), (long long)S, (long long)L));
        HIP_TRY(ctx, hipMemcpyAsync(j_blues, ctx->staging.p, L * S * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    work.release(); out_t.release();
    return TARDIS_MC_OK;
}

int tardis_mc_formal_integral(TardisMcContext *ctx, double inner_temperature, const double *frequencies, int64_t n_frequencies,
                              const double *att_S_ul, const double *Jred_lu, const double *Jblue_lu, int64_t n_impact_parameters,
                              double *luminosity_densities, double *intensities_nu_p)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->have_geometry || !ctx->have_opacity)
        return fail(ctx, TARDIS_MC_ERR_STATE, "formal integral needs set_geometry and set_opacity");
    if (!frequencies || !att_S_ul || !Jred_lu || !Jblue_lu || !luminosity_densities || n_frequencies < 0 || n_impact_parameters < 2 ||
        n_frequencies > (1LL << 30) || n_impact_parameters > 65535)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid formal integral arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t S = (size_t)ctx->n_shells, L = (size_t)ctx->n_lines, n_nu = (size_t)n_frequencies, N = (size_t)n_impact_parameters;
    if (n_nu == 0) return TARDIS_MC_OK;
    DevBuf work;  // [exp_tau | att | jred | jblue | freqs | z | I | Lum] doubles, then sid / n_int ints
    const size_t n_d = 4 * S * L + n_nu + N * 2 * S + n_nu * N + n_nu;
    HIP_TRY(ctx, work.ensure(n_d * sizeof(double) + (N * 2 * S + N) * sizeof(int)));
    double *d_exp = work.as<double>(), *d_att = d_exp + S * L, *d_jred = d_att + S * L, *d_jblue = d_jred + S * L,
           *d_freq = d_jblue + S * L, *d_z = d_freq + n_nu, *d_I = d_z + N * 2 * S, *d_lum = d_I + n_nu * N;
    int *d_sid = reinterpret_cast<int *>(d_lum + n_nu), *d_nint = d_sid + N * 2 * S;
    HIP_TRY(ctx, hipMemcpyAsync(d_att, att_S_ul, S * L * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_jred, Jred_lu, S * L * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_jblue, Jblue_lu, S * L * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_freq, frequencies, n_nu * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    if (S * L > 0)
        hipLaunchKernelGGL(mc::fi_exp_tau_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->tau_t.as<double>(), (long long)(S * L), d_exp);
    hipLaunchKernelGGL(mc::fi_intersections_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, ctx->stream, (int)S,
                       ctx->r_inner.as<double>(), ctx->r_outer.as<double>(), ctx->t_exp, (int)N, d_z, d_sid, d_nint);
    std::vector<double> r_last(1);
    HIP_TRY(ctx, hipMemcpyAsync(r_last.data(), ctx->r_outer.as<double>() + (S - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    mc::FormalIntegralArgs a{};
    a.n_shells = (int)S; a.n_lines = (int)L; a.n_nu = (int)n_nu; a.N = (int)N;
    a.t_exp = ctx->t_exp; a.inner_temperature = inner_temperature; a.radius_max = r_last[0];
    a.sigma_thomson = 6.652458734e-25;  // SIGMA_THOMSON, transport/montecarlo/configuration/constants.py:3 (astropy const13)
    a.r_inner = ctx->r_inner.as<double>(); a.nu_line = ctx->nu_line.as<double>(); a.n_e = ctx->n_e.as<double>();
    a.exp_tau = d_exp; a.att_S_ul = d_att; a.Jred_lu = d_jred; a.Jblue_lu = d_jblue; a.frequencies = d_freq;
    a.z = d_z; a.sid = d_sid; a.n_int = d_nint; a.intensities_nu_p = d_I;
    HIP_TRY(ctx, ctx->counters.ensure(TARDIS_MC_N_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, TARDIS_MC_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    a.line_steps = ctx->counters.as<unsigned long long>();  // counters[0] (line visits) = resonances crossed by the rays
    hipLaunchKernelGGL(mc::fi_rays_kernel, dim3((unsigned)((n_nu + 63) / 64), (unsigned)N), dim3(64), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(mc::fi_trapezoid_kernel, dim3((unsigned)((n_nu + 63) / 64)), dim3(64), 0, ctx->stream, d_I, (int)n_nu, (int)N,
                       a.radius_max, d_lum);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = true;
    ctx->chunks_timed = 0;
    HIP_TRY(ctx, hipMemcpyAsync(luminosity_densities, d_lum, n_nu * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (intensities_nu_p) HIP_TRY(ctx, hipMemcpyAsync(intensities_nu_p, d_I, n_nu * N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    work.release();
    return TARDIS_MC_OK;
}

/* ---- multi-GPU -------------------------------------------------------------------------------------- */
int tardis_mc_comm_get_unique_id(uint8_t out_id[TARDIS_MC_UNIQUE_ID_BYTES])
{
    std::string err;
    if (!out_id) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!load_rccl(err)) return fail(nullptr, TARDIS_MC_ERR_COMM, "%s", err.c_str());
    Id128 id;
    memset(&id, 0, sizeof id);
    int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(nullptr, TARDIS_MC_ERR_COMM, "ncclGetUniqueId failed (%d)", r);
    memcpy(out_id, id.bytes, TARDIS_MC_UNIQUE_ID_BYTES);
    return TARDIS_MC_OK;
}

int tardis_mc_comm_init(TardisMcContext *ctx, int rank, int world_size, const uint8_t id[TARDIS_MC_UNIQUE_ID_BYTES])
{
    if (!ctx || !id || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(ctx, TARDIS_MC_ERR_INVALID_ARGUMENT, "invalid communicator arguments");
    std::string err;
    if (!load_rccl(err)) return fail(ctx, TARDIS_MC_ERR_COMM, "%s", err.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Id128 uid;
    memcpy(uid.bytes, id, TARDIS_MC_UNIQUE_ID_BYTES);
    int r = g_rccl.CommInitRank(&ctx->comm, world_size, uid, rank);
    if (r != 0) return fail(ctx, TARDIS_MC_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    ctx->rank = rank;
    ctx->world = world_size;
    return TARDIS_MC_OK;
}

int tardis_mc_allreduce_estimators(TardisMcContext *ctx)
{
    if (!ctx) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (!ctx->est_valid) return fail(ctx, TARDIS_MC_ERR_STATE, "no estimators allocated");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = reduce_estimator_copies(ctx);
    if (rc) return rc;
    if (!ctx->comm) {
        if (ctx->world <= 1) return TARDIS_MC_OK;  // single process, no communicator: nothing to reduce
        return fail(ctx, TARDIS_MC_ERR_STATE, "tardis_mc_comm_init has not been called");
    }
    EstLayout e = est_layout(ctx->est_S, ctx->est_L, ctx->est_G, ctx->est_copies);
    // ncclDouble = 8, ncclSum = 0; one in-place all-reduce over [J | nu_bar | v-hist | j_blue | Edotlu]
    int r = g_rccl.AllReduce(ctx->est.p, ctx->est.p, e.reduce_elems, 8, 0, ctx->comm, ctx->stream);
    if (r != 0) return fail(ctx, TARDIS_MC_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return TARDIS_MC_OK;
}

// One-element all-reduce of (rank + 1): every rank must read N (N + 1) / 2 back -- the communicator really spans N ranks and sums.
__global__ void comm_check_fill_kernel(double *p, double v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

int tardis_mc_comm_check(TardisMcContext *ctx, int *out_ranks)
{
    if (!ctx || !out_ranks) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    *out_ranks = 0;
    if (!ctx->comm) return fail(ctx, TARDIS_MC_ERR_STATE, "tardis_mc_comm_init has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf cell;
    HIP_TRY(ctx, cell.ensure(sizeof(double)));
    hipLaunchKernelGGL(comm_check_fill_kernel, dim3(1), dim3(64), 0, ctx->stream, cell.as<double>(), (double)(ctx->rank + 1));
    hipError_t e = hipGetLastError();
    int r = e == hipSuccess ? g_rccl.AllReduce(cell.p, cell.p, 1, 8, 0, ctx->comm, ctx->stream) : 0;
    double got = 0.0;
    if (e == hipSuccess && r == 0) e = hipMemcpyAsync(&got, cell.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && r == 0) e = hipStreamSynchronize(ctx->stream);
    cell.release();
    HIP_TRY(ctx, e);
    if (r != 0) return fail(ctx, TARDIS_MC_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    const double want = 0.5 * (double)ctx->world * (double)(ctx->world + 1);
    if (got != want)
        return fail(ctx, TARDIS_MC_ERR_COMM, "communicator self-check: the sum of (rank + 1) over %d ranks came back as %.17g, not %.17g", ctx->world, got, want);
    *out_ranks = ctx->world;
    return TARDIS_MC_OK;
}

/* ---- diagnostics (numerics parity tests) ------------------------------------------------------------- */
int tardis_mc_debug_eval(TardisMcContext *ctx, int op, const double *x, const double *y, double *out, int64_t n)
{
    if (!ctx || !x || !out || n <= 0) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dx, dy, dout, scratch;
    size_t nx = (op == 7) ? 1 : (size_t)n;
    HIP_TRY(ctx, dx.ensure(nx * 8));
    HIP_TRY(ctx, dout.ensure((size_t)n * 8));
    HIP_TRY(ctx, scratch.ensure(mc::MT_N * 4));
    HIP_TRY(ctx, hipMemcpy(dx.p, x, nx * 8, hipMemcpyHostToDevice));
    if (y) { HIP_TRY(ctx, dy.ensure((size_t)n * 8)); HIP_TRY(ctx, hipMemcpy(dy.p, y, (size_t)n * 8, hipMemcpyHostToDevice)); }
    int blocks = op == 7 ? 1 : (int)((n + 255) / 256);
    hipLaunchKernelGGL(debug_eval_kernel, dim3(blocks), dim3(op == 7 ? 64 : 256), 0, ctx->stream, op, dx.as<double>(),
                       y ? dy.as<double>() : nullptr, dout.as<double>(), (long long)n, scratch.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    dx.release(); dy.release(); dout.release(); scratch.release();
    return TARDIS_MC_OK;
}

int tardis_mc_debug_microbench(TardisMcContext *ctx, int which, int64_t n_doubles, int iters, int blocks, double *out_ms)
{
    if (!ctx || !out_ms || n_doubles < 1024 || iters < 1 || blocks < 1) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf table, sink;
    HIP_TRY(ctx, table.ensure((size_t)n_doubles * 8));
    HIP_TRY(ctx, sink.ensure(8));
    HIP_TRY(ctx, hipMemsetAsync(table.p, 0, (size_t)n_doubles * 8, ctx->stream));
    for (int rep = 0; rep < 2; ++rep) {  // first launch warms up
            HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        if (which == 15) hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, table.as<double>(), (long long)n_doubles, iters);
        else hipLaunchKernelGGL(microbench_kernel, dim3(blocks), dim3(256), 0, ctx->stream, which, table.as<double>(),
                                (long long)n_doubles, iters, sink.as<double>());
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    }
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    *out_ms = ms;
    table.release(); sink.release();
    return TARDIS_MC_OK;
}

}  // extern "C"
