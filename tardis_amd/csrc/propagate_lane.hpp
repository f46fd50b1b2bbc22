// propagate_lane.hpp -- "lane per packet" propagation kernel (variant 0).
//
// One packet per lane, the whole reference event loop (classic/packet_propagation.py:52-251) executed by that
// lane.  Simple and bit-exact; used as the correctness baseline for the cooperative variants and for the
// feature paths that are not hot (v-packets, full relativity).  J / nu_bar go through an LDS-privatised
// per-workgroup accumulator (2*S doubles) flushed once per workgroup; j_blue / Edotlu are fp64 global atomics
// into shell-major tables.
#pragma once
#include "mc_device.hpp"

namespace mc {

struct LaneCounters {
    unsigned long long visits = 0, events = 0, macro = 0, vvisits = 0, vpackets = 0;
};

struct Tracker {
    double radius, nu, energy, before_nu, before_mu, before_energy, after_nu, after_mu, after_energy;
    long long shell_id, interaction_type, line_absorb_id, line_emit_id, interactions_count, boundary_buffer;
    __device__ void init()
    {
        const double nan = __builtin_nan("");
        radius = nu = energy = before_nu = before_mu = before_energy = after_nu = after_mu = after_energy = nan;
        shell_id = -1; interaction_type = -1; line_absorb_id = -1; line_emit_id = -1;
        interactions_count = 0; boundary_buffer = -1;
    }
    __device__ long long pop() { long long v = boundary_buffer; boundary_buffer = 0; return v; }
};

// ---- trace_packet (modes/homologous_rad_packet_transport.py:30-174)
template <bool FULL>
__device__ int trace_packet(const DeviceProblem &P, Packet &p, Rng &rng, double chi_cont, double *jb, double *ed,
                            double &distance, int &type, int &delta_shell, LaneCounters &cn)
{
    const int L = P.n_lines;
    const double t = P.t_exp;
    double d_boundary;
    distance_boundary(p.r, p.mu, P.r_inner[p.shell], P.r_outer[p.shell], d_boundary, delta_shell);
    const int start = p.next_line_id;
    double tau_event = -mcm::log(rng.random());
    double tau_lines = 0.0;
    double velocity = p.r / t;
    double dop = doppler_factor<FULL>(velocity, p.mu);
    double comov_nu = p.nu * dop;
    double d_cont = tau_event / chi_cont;
    const int last = L - 1;
    const double *tau_row = P.tau_t + (size_t)p.shell * L;
    double *jb_row = jb + (size_t)p.shell * L;
    double *ed_row = ed + (size_t)p.shell * L;
    // energy Doppler term of update_estimators_line, hoisted: (d + mu r) / (t c)
    const double mur = p.mu * p.r;
    const double tc = t * C_LIGHT;
    bool broke = false;
    distance = 0.0;
    type = 0;
    cn.events++;
    for (int cur = start; cur < L; ++cur) {
        cn.visits++;
        double nu_line = P.nu_line[cur];
        double tau_line = tau_row[cur];
        tau_lines += tau_line;
        double d_trace;
        if (!distance_line<FULL>(p.nu, p.r, p.mu, comov_nu, cur == last, nu_line, t, d_trace)) return ERR_MONTECARLO;
        double tau_cont = chi_cont * d_trace;
        double tau_combined = tau_lines + tau_cont;
        distance = d_trace;  // Python min(d_trace, d_boundary, d_cont)
        if (d_boundary < distance) distance = d_boundary;
        if (d_cont < distance) distance = d_cont;
        if (d_trace != 0) {
            if (distance == d_boundary) { type = IT_BOUNDARY; p.next_line_id = cur; broke = true; break; }
            if (distance == d_cont) { type = IT_ESCATTERING; p.next_line_id = cur; broke = true; break; }
        }
        // update_estimators_line (estimators/radfield_estimator_calcs.py:128-164)
        double energy;
        if (!FULL) energy = p.energy * (1.0 - ((d_trace + mur) / tc));
        else energy = p.energy;
        if (!(P.debug_flags & 1)) {
            atomic_add_f64(&jb_row[cur], energy / p.nu);
            atomic_add_f64(&ed_row[cur], energy);
        }
        if (tau_combined > tau_event && !P.disable_line_scattering) {
            type = IT_LINE; p.next_line_id = cur; distance = d_trace; broke = true; break;
        }
        d_cont = (tau_event - tau_lines) / chi_cont;
    }
    if (!broke) {  // for-else: next_line_id untouched
        if (d_cont < d_boundary) { distance = d_cont; type = IT_ESCATTERING; }
        else { distance = d_boundary; type = IT_BOUNDARY; }
    }
    return 0;
}

// ---- move_r_packet + update_estimators_bulk (packets/movement.py:31-76; radfield_estimator_calcs.py:25-53)
template <bool FULL>
__device__ __forceinline__ void move_r_packet(const DeviceProblem &P, Packet &p, double distance, double *lds_J,
                                              double *lds_nubar)
{
    double velocity = p.r / P.t_exp;
    double dop = doppler_factor<FULL>(velocity, p.mu);
    double r = p.r;
    if (distance > 0.0) {
        double new_r = sqrt(r * r + distance * distance + 2.0 * r * distance * p.mu);
        p.mu = (p.mu * r + distance) / new_r;
        p.r = new_r;
        double comov_nu = p.nu * dop;
        double comov_energy = p.energy * dop;
        if (FULL) distance *= dop;
        if (!(P.debug_flags & 2)) {
            atomicAdd(&lds_J[p.shell], comov_energy * distance);
            atomicAdd(&lds_nubar[p.shell], comov_energy * distance * comov_nu);
        }
    }
}

template <bool FULL>
__device__ __forceinline__ void line_emission(const DeviceProblem &P, Packet &p, int emission_line_id)
{ // interaction_events.py:227-258
    double velocity = p.r / P.t_exp;
    double inv = inverse_doppler_factor<FULL>(velocity, p.mu);
    p.nu = P.nu_line[emission_line_id] * inv;
    p.next_line_id = emission_line_id + 1;
    if (FULL) p.mu = aberration_cmf_to_lf(p.r, P.t_exp, p.mu);
}

// macro_atom_interaction (macro_atom.py:52-104)
__device__ inline int macro_atom_interaction(const DeviceProblem &P, Rng &rng, int level, int shell, int &out_line,
                                             int &out_type, LaneCounters &cn)
{
    const double *prob_row = P.prob_t + (size_t)shell * P.n_trans;
    int ttype = 0, tid = -1;
    while (ttype >= 0) {
        double probability = 0.0;
        double event = rng.random();
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

template <bool FULL>
__device__ int line_scatter_event(const DeviceProblem &P, Packet &p, Rng &rng, LaneCounters &cn)
{ // interaction_event_callers.py:187-239, :31-91
    double velocity = p.r / P.t_exp;
    double old_dop = doppler_factor<FULL>(velocity, p.mu);
    p.mu = 2.0 * rng.random() - 1.0;
    double inv_new = inverse_doppler_factor<FULL>(velocity, p.mu);
    double comov_energy = p.energy * old_dop;
    p.energy = comov_energy * inv_new;
    if (P.line_interaction_type == 0) {
        line_emission<FULL>(P, p, p.next_line_id);
        return 0;
    }
    double comov_nu = p.nu * old_dop;
    p.nu = comov_nu * inv_new;
    int level = P.line2level[p.next_line_id];
    int emit, ttype;
    int err = macro_atom_interaction(P, rng, level, p.shell, emit, ttype, cn);
    if (err) return err;
    if (ttype != -1) return ERR_UNSUPPORTED;
    line_emission<FULL>(P, p, emit);
    return 0;
}

template <bool FULL>
__device__ __forceinline__ void thomson_scatter(const DeviceProblem &P, Packet &p, Rng &rng)
{ // interaction_events.py:184-217
    double velocity = p.r / P.t_exp;
    double old_dop = doppler_factor<FULL>(velocity, p.mu);
    double comov_nu = p.nu * old_dop;
    double comov_energy = p.energy * old_dop;
    p.mu = 2.0 * rng.random() - 1.0;
    double inv_new = inverse_doppler_factor<FULL>(velocity, p.mu);
    p.nu = comov_nu * inv_new;
    p.energy = comov_energy * inv_new;
    if (FULL) p.mu = aberration_cmf_to_lf(p.r, P.t_exp, p.mu);
}

// ---- v-packets (packets/virtual_packet.py:82-386)
template <bool FULL>
__device__ int trace_vpacket_within_shell(const DeviceProblem &P, Packet &v, double &tau_out, double &d_boundary,
                                          int &delta, LaneCounters &cn)
{
    const int L = P.n_lines;
    const double t = P.t_exp;
    distance_boundary(v.r, v.mu, P.r_inner[v.shell], P.r_outer[v.shell], d_boundary, delta);
    const int start = v.next_line_id;
    double chi_e = P.n_e[v.shell] * P.sigma_thomson;
    double velocity = v.r / t;
    double dop = doppler_factor<FULL>(velocity, v.mu);
    double comov_nu = v.nu * dop;
    double chi_cont = chi_e;
    if (FULL) chi_cont *= dop;
    double tau = chi_cont * d_boundary;
    const double *tau_row = P.tau_t + (size_t)v.shell * L;
    int cur = start;
    bool broke = false;
    for (cur = start; cur < L; ++cur) {
        cn.vvisits++;
        double d_line;
        if (!distance_line<FULL>(v.nu, v.r, v.mu, comov_nu, cur == L - 1, P.nu_line[cur], t, d_line)) return ERR_MONTECARLO;
        if (d_boundary <= d_line) { broke = true; break; }
        tau += tau_row[cur];
    }
    if (!broke) {
        cur = (start < L) ? L - 1 : start;
        if (cur == L - 1) cur += 1;
    }
    v.next_line_id = cur;
    tau_out = tau;
    return 0;
}

template <bool FULL>
__device__ int trace_vpacket(const DeviceProblem &P, Packet &v, Rng &rng, double &tau_out, LaneCounters &cn)
{
    double tau = 0.0;
    for (;;) {
        double tau_shell, d_boundary;
        int delta;
        int err = trace_vpacket_within_shell<FULL>(P, v, tau_shell, d_boundary, delta, cn);
        if (err) return err;
        tau += tau_shell;
        cross_shell(v.shell, v.status, delta, P.n_shells);
        if (tau > P.tau_russian) {
            double ev = rng.random();
            if (ev > P.survival_probability) {
                v.energy = 0.0;
                v.status = ST_EMITTED;
            } else {
                v.energy = v.energy / P.survival_probability * mcm::exp(-tau);
                tau = 0.0;
            }
        }
        double new_r = sqrt(v.r * v.r + d_boundary * d_boundary + 2.0 * v.r * d_boundary * v.mu);
        v.mu = (v.mu * v.r + d_boundary) / new_r;
        v.r = new_r;
        if (v.status == ST_EMITTED) break;
    }
    tau_out = tau;
    return 0;
}

template <bool FULL>
__device__ int trace_vpacket_volley(const DeviceProblem &P, const Packet &p, Rng &rng, long long packet_index,
                                    int &vseq, LaneCounters &cn)
{
    if (p.nu < P.spawn_start || p.nu > P.spawn_end) return 0;
    const long long n_v = P.n_vpackets;
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
    double mu_bin = (1.0 - mu_min) / (double)n_v;
    double r_velocity = p.r / t;
    double r_dop = doppler_factor<FULL>(r_velocity, p.mu);
    for (long long i = 0; i < n_v; ++i) {
        double v_mu = mu_min + (double)i * mu_bin + rng.random() * mu_bin;
        double weight;
        if (on_inner) {
            if (!FULL) weight = 2 * v_mu / (double)n_v;
            else weight = 2 * (v_mu + beta_inner) / (2 * beta_inner + 1) / (double)n_v;
        } else
            weight = (1 - mu_min) / (double)(2 * n_v);
        if (FULL) v_mu = aberration_cmf_to_lf(p.r, t, v_mu);
        double v_dop = doppler_factor<FULL>(r_velocity, v_mu);
        double ratio = r_dop / v_dop;
        Packet v;
        v.r = p.r; v.mu = v_mu; v.nu = p.nu * ratio; v.energy = p.energy * weight * ratio;
        v.shell = p.shell; v.next_line_id = p.next_line_id; v.status = ST_IN_PROCESS;
        double tau_v;
        cn.vpackets++;
        int err = trace_vpacket<FULL>(P, v, rng, tau_v, cn);
        if (err) return err;
        v.energy *= mcm::exp(-tau_v);
        // add_vpacket_collection_to_histogram (modes/montecarlo_transport.py:166-195)
        if (!(v.nu < P.grid0 || v.nu > P.grid_last)) {
            long long idx = (long long)floor((v.nu - P.grid0) / P.delta_nu);
            atomic_add_f64(&P.vhist[idx], v.energy);
        }
        if (P.vlog_count) {
            unsigned long long slot = atomicAdd(P.vlog_count, 1ull);
            if ((long long)slot < P.vlog_capacity) {
                P.vlog_packet[slot] = packet_index; P.vlog_seq[slot] = vseq;
                P.vlog_nu[slot] = v.nu; P.vlog_energy[slot] = v.energy; P.vlog_mu[slot] = v_mu; P.vlog_r[slot] = p.r;
            }
        }
        ++vseq;
    }
    return 0;
}

// ---- packet_propagation (classic/packet_propagation.py:52-318)
template <bool FULL, bool VPK, bool TRACK>
__device__ int propagate_one(const DeviceProblem &P, long long i, Rng &rng, double *lds_J, double *lds_nubar, double *jb,
                             double *ed, LaneCounters &cn)
{
    const double t = P.t_exp;
    Packet p;
    p.r = P.r0[i]; p.mu = P.mu0[i]; p.nu = P.nu0[i]; p.energy = P.e0[i];
    p.shell = 0; p.status = ST_IN_PROCESS; p.next_line_id = 0;
    Tracker trk;
    if (TRACK) trk.init();
    int vseq = 0;
    int err;
    {   // set_packet_props_{partial,full}_relativity
        double velocity = p.r / t;
        double inv = inverse_doppler_factor<FULL>(velocity, p.mu);
        if (FULL) {
            double beta = (p.r / t) / C_LIGHT;
            p.nu *= inv;
            p.energy *= inv;
            p.mu = (p.mu + beta) / (1 + beta * p.mu);
        } else {
            p.nu *= inv;
            p.energy *= inv;
        }
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
    if (VPK) { if ((err = trace_vpacket_volley<FULL>(P, p, rng, i, vseq, cn))) return err; }
    if (TRACK) trk.boundary_buffer += 1;

    while (p.status == ST_IN_PROCESS) {
        double velocity = p.r / t;
        double dop = doppler_factor<FULL>(velocity, p.mu);
        double chi_e = P.n_e[p.shell] * P.sigma_thomson;
        if (FULL) chi_e *= dop;
        double distance;
        int type, delta;
        if ((err = trace_packet<FULL>(P, p, rng, chi_e, jb, ed, distance, type, delta, cn))) return err;
        move_r_packet<FULL>(P, p, distance, lds_J, lds_nubar);
        if (type == IT_BOUNDARY) {
            if (TRACK) trk.boundary_buffer += 1;
            cross_shell(p.shell, p.status, delta, P.n_shells);
        } else if (type == IT_LINE) {
            if (TRACK) {
                trk.before_nu = p.nu; trk.before_mu = p.mu; trk.before_energy = p.energy;
                trk.line_absorb_id = p.next_line_id;
            }
            if ((err = line_scatter_event<FULL>(P, p, rng, cn))) return err;
            if (TRACK) {
                trk.after_nu = p.nu; trk.after_mu = p.mu; trk.after_energy = p.energy;
                trk.line_emit_id = p.next_line_id - 1;
                trk.interactions_count += 1 + trk.pop();
                trk.radius = p.r; trk.nu = p.nu; trk.energy = p.energy; trk.shell_id = p.shell;
                trk.interaction_type = IT_LINE;
            }
        } else {  // IT_ESCATTERING
            if (TRACK) {
                trk.before_mu = p.mu; trk.before_nu = p.nu; trk.before_energy = p.energy;
                trk.line_absorb_id = -1; trk.line_emit_id = -1;
            }
            thomson_scatter<FULL>(P, p, rng);
            if (TRACK) {
                trk.after_mu = p.mu; trk.after_nu = p.nu; trk.after_energy = p.energy;
                trk.interactions_count += 1 + trk.pop();
                trk.radius = p.r; trk.nu = p.nu; trk.energy = p.energy; trk.shell_id = p.shell;
                trk.interaction_type = IT_ESCATTERING;
            }
        }
        // one shared call site for the volley after a line or electron-scattering interaction: lanes of both kinds trace
        // their v-packets together instead of serialising two inlined copies of the volley
        if (VPK && type != IT_BOUNDARY) { if ((err = trace_vpacket_volley<FULL>(P, p, rng, i, vseq, cn))) return err; }
    }
    // set_packet_collection_output (modes/montecarlo_transport.py:70-90)
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
    return 0;
}

__device__ __forceinline__ int xcc_id()
{
    // HW_REG_XCC_ID (id 20), bits [3:0]: the physical XCD this wave runs on (MI355X_MICROARCH.md §Workgroup dispatch)
    return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf);
}

template <bool FULL, bool VPK, bool TRACK>
__global__ void __launch_bounds__(256) propagate_lane_kernel(DeviceProblem P)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lds_J = lds, *lds_nubar = lds + P.n_shells;
    for (int s = threadIdx.x; s < 2 * P.n_shells; s += blockDim.x) lds[s] = 0.0;
    __syncthreads();

    const long long n_threads = (long long)gridDim.x * blockDim.x;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // per-XCD private line-estimator copy: plain L2 atomics never cross an XCD boundary
    const int copy = P.n_est_copies > 1 ? (xcc_id() % P.n_est_copies) : 0;
    double *jb = P.jblue_t + (size_t)copy * P.est_copy_stride;
    double *ed = P.edot_t + (size_t)copy * P.est_copy_stride;
    Rng rng;
    uint32_t *state = P.rng_state + (size_t)gtid * MT_N;
    LaneCounters cn;
    unsigned long long draws = 0;
    for (long long i = gtid; i < P.n_packets; i += n_threads) {
        rng.seed(state, P.seeds[i]);
        int err = propagate_one<FULL, VPK, TRACK>(P, i, rng, lds_J, lds_nubar, jb, ed, cn);
        draws += (unsigned long long)rng.draws;
        if (err) {
            long long prev = atomicMin(&P.first_error[0], i);
            if (i < prev) P.first_error[1] = err;  // best effort; the host re-derives the code of the min packet
            P.out_nu[i] = (double)err;             // marker consumed by the host
            P.out_e[i] = -99.0;
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < P.n_shells; s += blockDim.x) {
        if (lds_J[s] != 0.0) atomic_add_f64(&P.J[s], lds_J[s]);
        if (lds_nubar[s] != 0.0) atomic_add_f64(&P.nubar[s], lds_nubar[s]);
    }
    // counters: one atomic per lane is fine (once per thread lifetime)
    atomicAdd(&P.counters[0], cn.visits);
    atomicAdd(&P.counters[1], cn.events);
    atomicAdd(&P.counters[2], cn.macro);
    atomicAdd(&P.counters[3], cn.vvisits);
    atomicAdd(&P.counters[4], cn.vpackets);
    atomicAdd(&P.counters[5], draws);
}

}  // namespace mc
