/* TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the reference's Monte Carlo packet-propagation path.
 *
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  It restates, in plain scalar C with the reference's operation order (IEEE double, no FMA
 * contraction: build with -ffp-contract=off), the algorithm of
 *
 *   montecarlo_transport_with_vpackets   tardis/transport/montecarlo/modes/montecarlo_transport.py:238-373
 *   packet_propagation                   tardis/transport/montecarlo/modes/classic/packet_propagation.py:52-318
 *   trace_packet                         tardis/transport/montecarlo/modes/homologous_rad_packet_transport.py:30-174
 *   calculate_distance_*                 tardis/transport/geometry/calculate_distances.py:25-112,198-219
 *   Doppler / aberration                 tardis/transport/frame_transformations.py:12-109
 *   move_r_packet / shell crossing       tardis/transport/montecarlo/packets/movement.py:31-102
 *   estimator updates                    tardis/transport/montecarlo/estimators/radfield_estimator_calcs.py:25-53,128-164
 *   thomson_scatter / line_emission      tardis/transport/montecarlo/interaction_events.py:184-258
 *   line_scatter_event / macro_atom_event tardis/transport/montecarlo/interaction_event_callers.py:31-91,187-239
 *   macro_atom_interaction               tardis/transport/montecarlo/macro_atom.py:52-104
 *   v-packets                            tardis/transport/montecarlo/packets/virtual_packet.py:82-386
 *   TrackerLastInteraction               tardis/transport/montecarlo/packets/trackers/tracker_last_interaction.py:8-254
 *   RNG: numpy legacy MT19937 (np.random.seed / np.random.random inside njit; montecarlo_transport.py:65)
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against golden vectors produced by the
 * reference itself (run in pure-Python mode in the dev container, tools/make_golden.py) and against the
 * hard-coded known answers of the reference's own tests (tests/test_oracle_kat.py).
 *
 * math_mode 0: libm log/exp (what the reference calls).  math_mode 1: portable_math.h (bit-identical to
 * the HIP kernels' implementation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/tardis_mc.h"
#include "portable_math.h"

#define C_SPEED_OF_LIGHT 2.99792458e10 /* tardis/constants.py:1 (CODATA 2010) */
#define CLOSE_LINE_THRESHOLD 1e-14     /* configuration/constants.py:4 */
#define MISS_DISTANCE 1e99             /* configuration/constants.py:6 */

enum { IT_BOUNDARY = 1, IT_LINE = 2, IT_ESCATTERING = 4 };           /* radiative_packet.py:12-36 */
enum { ST_IN_PROCESS = 0, ST_EMITTED = 1, ST_REABSORBED = 2 };       /* radiative_packet.py:39-43 */

/* ------------------------------------------------------------------ MT19937 (numpy legacy stream) -- */
typedef struct { uint32_t mt[624]; int idx; int64_t draws; } mt_state;

static void mt_seed(mt_state *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
    s->draws = 0;
}

static uint32_t mt_u32(mt_state *s)
{
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static double mt_random(mt_state *s)
{
    uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    s->draws++;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* ------------------------------------------------------------------------------- run-wide context -- */
typedef struct {
    const TardisMcGeometry *geo;
    const TardisMcOpacity *op;
    const TardisMcConfig *cfg;
    int math_mode;
    int atomic; /* >1 thread: estimator updates use omp atomic */
    double *J, *nubar, *jblue, *edot;
} run_ctx;

typedef struct {
    double r, mu, nu, energy;
    int64_t next_line_id, current_shell_id, status;
} rpacket;

typedef struct {
    double radius, nu, energy, before_nu, before_mu, before_energy, after_nu, after_mu, after_energy;
    int64_t shell_id, interaction_type, line_absorb_id, line_emit_id, interactions_count, boundary_buffer;
} tracker;

typedef struct { /* growable per-packet v-packet list (VPacketCollection) */
    double *nus, *energies, *mus, *rs;
    int64_t n, cap;
} vlist;

typedef struct { int64_t c[TARDIS_MC_N_COUNTERS]; } counters;

static double m_log(const run_ctx *c, double x) { return c->math_mode ? pm_log(x) : log(x); }
static double m_exp(const run_ctx *c, double x) { return c->math_mode ? pm_exp(x) : exp(x); }

static void add_to(const run_ctx *c, double *p, double v)
{
    if (c->atomic) {
#pragma omp atomic
        *p += v;
    } else
        *p += v;
}

/* ------------------------------------------------------------------------- frame transformations -- */
static double doppler_factor(double velocity, double mu, int full)
{ /* frame_transformations.py:12-40 */
    double inv_c = 1 / C_SPEED_OF_LIGHT;
    double beta = velocity * inv_c;
    if (!full) return 1.0 - mu * beta;
    return (1.0 - mu * beta) / sqrt(1 - beta * beta);
}

static double inverse_doppler_factor(double velocity, double mu, int full)
{ /* frame_transformations.py:43-70 */
    double inv_c = 1 / C_SPEED_OF_LIGHT;
    double beta = velocity * inv_c;
    if (!full) return 1.0 / (1.0 - mu * beta);
    return (1.0 + mu * beta) / sqrt(1 - beta * beta);
}

static double aberration_cmf_to_lf(double r, double t, double mu)
{ /* frame_transformations.py:89-97 */
    double ct = C_SPEED_OF_LIGHT * t;
    double beta = r / ct;
    return (mu + beta) / (1.0 + beta * mu);
}

static double aberration_lf_to_cmf(double r, double t, double mu)
{ /* frame_transformations.py:100-109 */
    double ct = C_SPEED_OF_LIGHT * t;
    double beta = r / ct;
    return (mu - beta) / (1.0 - beta * mu);
}

/* --------------------------------------------------------------------------------------- distances -- */
static void distance_boundary(double r, double mu, double r_inner, double r_outer, double *d, int64_t *delta)
{ /* calculate_distances.py:25-62 */
    if (mu > 0.0) {
        *d = sqrt(r_outer * r_outer + ((mu * mu - 1.0) * r * r)) - (r * mu);
        *delta = 1;
    } else {
        double check = r_inner * r_inner + (r * r * (mu * mu - 1.0));
        if (check >= 0.0) {
            *d = -r * mu - sqrt(check);
            *delta = -1;
        } else {
            *d = sqrt(r_outer * r_outer + ((mu * mu - 1.0) * r * r)) - (r * mu);
            *delta = 1;
        }
    }
}

static double distance_line_full_relativity(double nu_line, double nu, double t, double r, double mu)
{ /* calculate_distances.py:198-219 */
    double nu_r = nu_line / nu;
    double ct = C_SPEED_OF_LIGHT * t;
    return -mu * r + (ct - nu_r * nu_r * sqrt(ct * ct - (1 + r * r * (1 - mu * mu) * (1 + 1.0 / (nu_r * nu_r))))) /
                         (1 + nu_r * nu_r);
}

/* returns 0 or TARDIS_MC_ERR_MONTECARLO */
static int distance_line(double nu, double r, double mu, double comov_nu, int is_last_line, double nu_line, double t,
                         int full, double *d)
{ /* calculate_distances.py:66-112 */
    if (is_last_line) { *d = MISS_DISTANCE; return 0; }
    double nu_diff = comov_nu - nu_line;
    if (fabs(nu_diff / nu) < CLOSE_LINE_THRESHOLD) { *d = 0.0; return 0; }
    double distance;
    if (nu_diff >= 0) distance = (nu_diff / nu) * C_SPEED_OF_LIGHT * t;
    else return TARDIS_MC_ERR_MONTECARLO;
    if (full) { *d = distance_line_full_relativity(nu_line, nu, t, r, mu); return 0; }
    *d = distance;
    return 0;
}

/* --------------------------------------------------------------------------------------- estimators -- */
static void update_estimators_line(const run_ctx *c, const rpacket *p, int64_t line, double d_trace, double t, int full)
{ /* radfield_estimator_calcs.py:128-164 ; frame_transformations.py:73-85 */
    double energy;
    if (!full) {
        double dop = 1.0 - ((d_trace + p->mu * p->r) / (t * C_SPEED_OF_LIGHT));
        energy = p->energy * dop;
    } else
        energy = p->energy;
    int64_t S = c->op->n_shells;
    add_to(c, &c->jblue[line * S + p->current_shell_id], energy / p->nu);
    add_to(c, &c->edot[line * S + p->current_shell_id], energy);
}

static void move_r_packet(const run_ctx *c, rpacket *p, double distance, int full)
{ /* movement.py:31-76 ; radfield_estimator_calcs.py:25-53 */
    double velocity = p->r / c->geo->time_explosion;
    double dop = doppler_factor(velocity, p->mu, full);
    double r = p->r;
    if (distance > 0.0) {
        double new_r = sqrt(r * r + distance * distance + 2.0 * r * distance * p->mu);
        p->mu = (p->mu * r + distance) / new_r;
        p->r = new_r;
        double comov_nu = p->nu * dop;
        double comov_energy = p->energy * dop;
        if (full) distance *= dop;
        add_to(c, &c->J[p->current_shell_id], comov_energy * distance);
        add_to(c, &c->nubar[p->current_shell_id], comov_energy * distance * comov_nu);
    }
}

static void move_across_shell_boundary(int64_t *shell, int64_t *status, int64_t delta, int64_t n_shells)
{ /* movement.py:80-102 */
    int64_t next = *shell + delta;
    if (next >= n_shells) *status = ST_EMITTED;
    else if (next < 0) *status = ST_REABSORBED;
    else *shell = next;
}

/* ------------------------------------------------------------------------------------ trace_packet -- */
/* Analysis hook (tools/locality_stats.py; single-threaded runs only): every trace's {shell, first line, lines visited, event
 * type} appended to a caller-owned buffer.  Not part of any parity path. */
static int64_t *g_trace_log = NULL, g_trace_log_cap = 0, g_trace_log_n = 0;
void oracle_set_trace_log(int64_t *buf, int64_t capacity_records) { g_trace_log = buf; g_trace_log_cap = capacity_records; g_trace_log_n = 0; }
int64_t oracle_trace_log_count(void) { return g_trace_log_n; }
/* analysis hook (tools/vpacket_stats.py), single-threaded runs only: one record {first shell, shell crossings traced, lines visited,
 * crossing at which the Russian roulette dropped it or -1} per v-packet */
static int64_t *g_vtrace_log = NULL, g_vtrace_log_cap = 0, g_vtrace_log_n = 0;
void oracle_set_vtrace_log(int64_t *buf, int64_t capacity_records) { g_vtrace_log = buf; g_vtrace_log_cap = capacity_records; g_vtrace_log_n = 0; }
int64_t oracle_vtrace_log_count(void) { return g_vtrace_log_n; }

static int trace_packet(const run_ctx *c, rpacket *p, mt_state *rng, double chi_cont, double *out_distance,
                        int *out_type, int64_t *out_delta, counters *cn)
{ /* homologous_rad_packet_transport.py:30-174 (continuum_process_enabled = False, escat_prob = 1) */
    const TardisMcOpacity *op = c->op;
    const int full = c->cfg->enable_full_relativity;
    const double t = c->geo->time_explosion;
    const int64_t L = op->n_lines, S = op->n_shells;
    double r_inner = c->geo->r_inner[p->current_shell_id];
    double r_outer = c->geo->r_outer[p->current_shell_id];
    double d_boundary;
    int64_t delta_shell;
    distance_boundary(p->r, p->mu, r_inner, r_outer, &d_boundary, &delta_shell);

    int64_t start = p->next_line_id;
    double tau_event = -m_log(c, mt_random(rng));
    double tau_lines = 0.0;
    double velocity = p->r / t;
    double dop = doppler_factor(velocity, p->mu, full);
    double comov_nu = p->nu * dop;
    double d_cont = tau_event / chi_cont;
    int64_t last = L - 1;
    double distance = 0.0;
    int type = 0, broke = 0;
    cn->c[TARDIS_MC_CNT_EVENTS]++;

    for (int64_t cur = start; cur < L; ++cur) {
        cn->c[TARDIS_MC_CNT_LINE_VISITS]++;
        double nu_line = op->line_list_nu[cur];
        double tau_line = op->tau_sobolev[cur * S + p->current_shell_id];
        tau_lines += tau_line;
        double d_trace;
        int err = distance_line(p->nu, p->r, p->mu, comov_nu, cur == last, nu_line, t, full, &d_trace);
        if (err) return err;
        double tau_cont = chi_cont * d_trace;
        double tau_combined = tau_lines + tau_cont;
        /* Python min(a, b, c) */
        distance = d_trace;
        if (d_boundary < distance) distance = d_boundary;
        if (d_cont < distance) distance = d_cont;
        if (d_trace != 0) {
            if (distance == d_boundary) { type = IT_BOUNDARY; p->next_line_id = cur; broke = 1; break; }
            if (distance == d_cont) { type = IT_ESCATTERING; p->next_line_id = cur; broke = 1; break; }
        }
        update_estimators_line(c, p, cur, d_trace, t, full);
        if (tau_combined > tau_event && !c->cfg->disable_line_scattering) {
            type = IT_LINE; p->next_line_id = cur; distance = d_trace; broke = 1; break;
        }
        d_cont = (tau_event - tau_lines) / chi_cont;
    }
    if (!broke) { /* for-else (lines 157-172): next_line_id is left untouched */
        if (d_cont < d_boundary) { distance = d_cont; type = IT_ESCATTERING; }
        else { distance = d_boundary; type = IT_BOUNDARY; }
    }
    if (g_trace_log && g_trace_log_n < g_trace_log_cap) {
        int64_t *r = g_trace_log + 4 * g_trace_log_n++;
        r[0] = p->current_shell_id; r[1] = start; r[2] = (broke ? p->next_line_id + 1 : L) - start; r[3] = type;
    }
    *out_distance = distance;
    *out_type = type;
    *out_delta = delta_shell;
    return 0;
}

/* ------------------------------------------------------------------------------------ interactions -- */
static void line_emission(const run_ctx *c, rpacket *p, int64_t emission_line_id)
{ /* interaction_events.py:227-258 */
    const int full = c->cfg->enable_full_relativity;
    double t = c->geo->time_explosion;
    double velocity = p->r / t;
    double inv = inverse_doppler_factor(velocity, p->mu, full);
    p->nu = c->op->line_list_nu[emission_line_id] * inv;
    p->next_line_id = emission_line_id + 1;
    if (full) p->mu = aberration_cmf_to_lf(p->r, t, p->mu);
}

static int macro_atom_interaction(const run_ctx *c, mt_state *rng, int64_t level, int64_t shell, int64_t *out_line,
                                  int64_t *out_type, counters *cn)
{ /* macro_atom.py:52-104 */
    const TardisMcOpacity *op = c->op;
    int64_t S = op->n_shells, ttype = 0, tid = -1;
    while (ttype >= 0) {
        double probability = 0.0;
        double event = mt_random(rng);
        int64_t b0 = op->macro_block_edge_index[level], b1 = op->macro_block_edge_index[level + 1];
        int found = 0;
        for (tid = b0; tid < b1; ++tid) {
            cn->c[TARDIS_MC_CNT_MACRO_TRANSITIONS]++;
            probability += op->transition_probabilities[tid * S + shell];
            if (probability > event) {
                level = op->destination_level_id[tid];
                ttype = op->transition_type[tid];
                found = 1;
                break;
            }
        }
        if (!found) return TARDIS_MC_ERR_MACRO_ATOM;
    }
    *out_line = op->transition_line_id[tid];
    *out_type = ttype;
    return 0;
}

static int line_scatter_event(const run_ctx *c, rpacket *p, mt_state *rng, counters *cn)
{ /* interaction_event_callers.py:187-239 and :31-91 (classic mode: only BB_EMISSION = -1 is reachable) */
    const int full = c->cfg->enable_full_relativity;
    double t = c->geo->time_explosion;
    double velocity = p->r / t;
    double old_dop = doppler_factor(velocity, p->mu, full);
    p->mu = 2.0 * mt_random(rng) - 1.0; /* utils.py:14-15 */
    double inv_new = inverse_doppler_factor(velocity, p->mu, full);
    double comov_energy = p->energy * old_dop;
    p->energy = comov_energy * inv_new;
    if (c->cfg->line_interaction_type == TARDIS_MC_LINE_SCATTER) {
        line_emission(c, p, p->next_line_id);
        return 0;
    }
    double comov_nu = p->nu * old_dop;
    p->nu = comov_nu * inv_new;
    int64_t level = c->op->line2macro_level_upper[p->next_line_id];
    int64_t emit, ttype;
    int err = macro_atom_interaction(c, rng, level, p->current_shell_id, &emit, &ttype, cn);
    if (err) return err;
    if (ttype != -1) return TARDIS_MC_ERR_UNSUPPORTED; /* bf/ff/adiabatic channels are IIP-only */
    line_emission(c, p, emit);
    return 0;
}

static void thomson_scatter(const run_ctx *c, rpacket *p, mt_state *rng)
{ /* interaction_events.py:184-217 */
    const int full = c->cfg->enable_full_relativity;
    double t = c->geo->time_explosion;
    double velocity = p->r / t;
    double old_dop = doppler_factor(velocity, p->mu, full);
    double comov_nu = p->nu * old_dop;
    double comov_energy = p->energy * old_dop;
    p->mu = 2.0 * mt_random(rng) - 1.0;
    double inv_new = inverse_doppler_factor(velocity, p->mu, full);
    p->nu = comov_nu * inv_new;
    p->energy = comov_energy * inv_new;
    if (full) p->mu = aberration_cmf_to_lf(p->r, t, p->mu);
}

/* --------------------------------------------------------------------------------------- v-packets -- */
typedef struct { double r, mu, nu, energy; int64_t next_line_id, current_shell_id, status; } vpacket;

static int trace_vpacket_within_shell(const run_ctx *c, vpacket *v, double *tau_out, double *d_boundary_out,
                                      int64_t *delta_out, counters *cn)
{ /* virtual_packet.py:82-175 */
    const TardisMcOpacity *op = c->op;
    const int full = c->cfg->enable_full_relativity;
    const double t = c->geo->time_explosion;
    const int64_t L = op->n_lines, S = op->n_shells;
    double d_boundary;
    int64_t delta;
    distance_boundary(v->r, v->mu, c->geo->r_inner[v->current_shell_id], c->geo->r_outer[v->current_shell_id],
                      &d_boundary, &delta);
    int64_t start = v->next_line_id;
    double chi_e = op->electron_density[v->current_shell_id] * c->cfg->sigma_thomson;
    double velocity = v->r / t;
    double dop = doppler_factor(velocity, v->mu, full);
    double comov_nu = v->nu * dop;
    double chi_cont = chi_e;
    if (full) chi_cont *= dop;
    double tau = chi_cont * d_boundary;
    int64_t cur = start;
    int broke = 0;
    for (cur = start; cur < L; ++cur) {
        cn->c[TARDIS_MC_CNT_VPACKET_LINE_VISITS]++;
        double nu_line = op->line_list_nu[cur];
        double tau_line = op->tau_sobolev[cur * S + v->current_shell_id];
        double d_line;
        int err = distance_line(v->nu, v->r, v->mu, comov_nu, cur == L - 1, nu_line, t, full, &d_line);
        if (err) return err;
        if (d_boundary <= d_line) { broke = 1; break; }
        tau += tau_line;
    }
    if (!broke) { /* for-else: Python leaves cur at the last iterated value (or start if the range was empty) */
        cur = (start < L) ? L - 1 : start;
        if (cur == L - 1) cur += 1;
    }
    v->next_line_id = cur;
    *tau_out = tau;
    *d_boundary_out = d_boundary;
    *delta_out = delta;
    return 0;
}

static int trace_vpacket(const run_ctx *c, vpacket *v, mt_state *rng, double *tau_out, counters *cn)
{ /* virtual_packet.py:179-244 */
    double tau = 0.0;
    const double tau_russian = c->cfg->vpacket_tau_russian, survival = c->cfg->survival_probability;
    const int64_t log_shell = v->current_shell_id, log_visits0 = cn->c[TARDIS_MC_CNT_VPACKET_LINE_VISITS];
    int64_t log_crossings = 0, log_dropped = -1;
    for (;;) {
        double tau_shell, d_boundary;
        int64_t delta;
        int err = trace_vpacket_within_shell(c, v, &tau_shell, &d_boundary, &delta, cn);
        if (err) return err;
        ++log_crossings;
        tau += tau_shell;
        move_across_shell_boundary(&v->current_shell_id, &v->status, delta, c->geo->n_shells);
        if (tau > tau_russian) {
            double ev = mt_random(rng);
            if (ev > survival) {
                v->energy = 0.0;
                v->status = ST_EMITTED;
                log_dropped = log_crossings;
            } else {
                v->energy = v->energy / survival * m_exp(c, -tau);
                tau = 0.0;
            }
        }
        double new_r = sqrt(v->r * v->r + d_boundary * d_boundary + 2.0 * v->r * d_boundary * v->mu);
        v->mu = (v->mu * v->r + d_boundary) / new_r;
        v->r = new_r;
        if (v->status == ST_EMITTED) break;
    }
    if (g_vtrace_log && g_vtrace_log_n < g_vtrace_log_cap) {
        int64_t *r = g_vtrace_log + 4 * g_vtrace_log_n++;
        r[0] = log_shell; r[1] = log_crossings; r[2] = cn->c[TARDIS_MC_CNT_VPACKET_LINE_VISITS] - log_visits0; r[3] = log_dropped;
    }
    *tau_out = tau;
    return 0;
}

static void vlist_add(vlist *l, int64_t n_v, double nu, double energy, double mu, double r)
{ /* packet_collections.py:188-273 (growth rule: 2*len + n_v) */
    if (l->n >= l->cap) {
        int64_t ncap = l->cap * 2 + n_v;
        if (ncap < 4) ncap = 4;
        l->nus = (double *)realloc(l->nus, ncap * sizeof(double));
        l->energies = (double *)realloc(l->energies, ncap * sizeof(double));
        l->mus = (double *)realloc(l->mus, ncap * sizeof(double));
        l->rs = (double *)realloc(l->rs, ncap * sizeof(double));
        l->cap = ncap;
    }
    l->nus[l->n] = nu; l->energies[l->n] = energy; l->mus[l->n] = mu; l->rs[l->n] = r;
    l->n++;
}

static int trace_vpacket_volley(const run_ctx *c, const rpacket *p, mt_state *rng, vlist *vl, counters *cn)
{ /* virtual_packet.py:248-386 */
    const TardisMcConfig *cfg = c->cfg;
    const int full = cfg->enable_full_relativity;
    const double t = c->geo->time_explosion;
    if (p->nu < cfg->vpacket_spawn_start_frequency || p->nu > cfg->vpacket_spawn_end_frequency) return 0;
    int64_t n_v = cfg->number_of_vpackets;
    if (n_v == 0) return 0;
    double mu_min, beta_inner = 0.0;
    int on_inner;
    if (p->r > c->geo->r_inner[0]) {
        double r_inner_over_r = c->geo->r_inner[0] / p->r;
        mu_min = -sqrt(1 - r_inner_over_r * r_inner_over_r);
        on_inner = 0;
        if (full) mu_min = aberration_lf_to_cmf(p->r, t, mu_min);
    } else {
        on_inner = 1;
        mu_min = 0.0;
        if (full) {
            double inv_c = 1 / C_SPEED_OF_LIGHT;
            double inv_t = 1 / t;
            beta_inner = c->geo->r_inner[0] * inv_t * inv_c;
        }
    }
    double mu_bin = (1.0 - mu_min) / n_v;
    double r_velocity = p->r / t;
    double r_dop = doppler_factor(r_velocity, p->mu, full);
    for (int64_t i = 0; i < n_v; ++i) {
        double v_mu = mu_min + i * mu_bin + mt_random(rng) * mu_bin;
        double weight;
        if (on_inner) {
            if (!full) weight = 2 * v_mu / n_v;
            else weight = 2 * (v_mu + beta_inner) / (2 * beta_inner + 1) / n_v;
        } else
            weight = (1 - mu_min) / (2 * n_v);
        if (full) v_mu = aberration_cmf_to_lf(p->r, t, v_mu);
        double v_dop = doppler_factor(r_velocity, v_mu, full);
        double ratio = r_dop / v_dop;
        vpacket v;
        v.r = p->r; v.mu = v_mu; v.nu = p->nu * ratio; v.energy = p->energy * weight * ratio;
        v.current_shell_id = p->current_shell_id; v.next_line_id = p->next_line_id; v.status = ST_IN_PROCESS;
        double tau_v;
        cn->c[TARDIS_MC_CNT_VPACKETS]++;
        int err = trace_vpacket(c, &v, rng, &tau_v, cn);
        if (err) return err;
        v.energy *= m_exp(c, -tau_v);
        vlist_add(vl, n_v, v.nu, v.energy, v_mu, p->r);
    }
    return 0;
}

/* --------------------------------------------------------------------------------- tracker helpers -- */
static void tracker_init(tracker *k)
{
    k->radius = k->nu = k->energy = NAN;
    k->before_nu = k->before_mu = k->before_energy = NAN;
    k->after_nu = k->after_mu = k->after_energy = NAN;
    k->shell_id = -1; k->interaction_type = -1; k->line_absorb_id = -1; k->line_emit_id = -1;
    k->interactions_count = 0; k->boundary_buffer = -1;
}
static int64_t tracker_pop(tracker *k) { int64_t v = k->boundary_buffer; k->boundary_buffer = 0; return v; }

/* ------------------------------------------------------------------------------ packet_propagation -- */
static int packet_propagation(const run_ctx *c, rpacket *p, mt_state *rng, tracker *trk, vlist *vl, counters *cn)
{ /* classic/packet_propagation.py:52-318 */
    const TardisMcConfig *cfg = c->cfg;
    const int full = cfg->enable_full_relativity;
    const double t = c->geo->time_explosion;
    int err;
    /* set_packet_props_{partial,full}_relativity (:254-318) */
    if (full) {
        double beta = (p->r / t) / C_SPEED_OF_LIGHT;
        double velocity = p->r / t;
        double inv = inverse_doppler_factor(velocity, p->mu, 1);
        p->nu *= inv;
        p->energy *= inv;
        p->mu = (p->mu + beta) / (1 + beta * p->mu);
    } else {
        double velocity = p->r / t;
        double inv = inverse_doppler_factor(velocity, p->mu, 0);
        p->nu *= inv;
        p->energy *= inv;
    }
    { /* RPacket.initialize_line_id (radiative_packet.py:96-110): L - searchsorted(nu[::-1], comov_nu, 'left') */
        const int64_t L = c->op->n_lines;
        double velocity = p->r / t;
        double comov_nu = p->nu * doppler_factor(velocity, p->mu, full);
        int64_t lo = 0, hi = L; /* count of lines with nu_line >= comov_nu in the descending list */
        while (lo < hi) {
            int64_t mid = (lo + hi) / 2;
            if (c->op->line_list_nu[mid] >= comov_nu) lo = mid + 1; else hi = mid;
        }
        int64_t next = lo;
        if (next == L) next -= 1;
        p->next_line_id = next;
    }
    if ((err = trace_vpacket_volley(c, p, rng, vl, cn))) return err;
    trk->boundary_buffer += 1; /* track_boundary_event(from -1 to 0) */

    while (p->status == ST_IN_PROCESS) {
        double velocity = p->r / c->geo->time_explosion;
        double dop = doppler_factor(velocity, p->mu, full);
        double chi_e = c->op->electron_density[p->current_shell_id] * cfg->sigma_thomson; /* opacities.py:49-67 */
        if (full) chi_e *= dop;
        double distance;
        int type;
        int64_t delta;
        if ((err = trace_packet(c, p, rng, chi_e, &distance, &type, &delta, cn))) return err;
        if (type == IT_BOUNDARY) {
            move_r_packet(c, p, distance, full);
            trk->boundary_buffer += 1;
            move_across_shell_boundary(&p->current_shell_id, &p->status, delta, c->geo->n_shells);
        } else if (type == IT_LINE) {
            move_r_packet(c, p, distance, full);
            trk->before_nu = p->nu; trk->before_mu = p->mu; trk->before_energy = p->energy;
            trk->line_absorb_id = p->next_line_id;
            if ((err = line_scatter_event(c, p, rng, cn))) return err;
            trk->after_nu = p->nu; trk->after_mu = p->mu; trk->after_energy = p->energy;
            trk->line_emit_id = p->next_line_id - 1;
            trk->interactions_count += 1 + tracker_pop(trk);
            trk->radius = p->r; trk->nu = p->nu; trk->energy = p->energy; trk->shell_id = p->current_shell_id;
            trk->interaction_type = IT_LINE;
            if ((err = trace_vpacket_volley(c, p, rng, vl, cn))) return err;
        } else if (type == IT_ESCATTERING) {
            move_r_packet(c, p, distance, full);
            trk->before_mu = p->mu; trk->before_nu = p->nu; trk->before_energy = p->energy;
            trk->line_absorb_id = -1; trk->line_emit_id = -1;
            thomson_scatter(c, p, rng);
            trk->after_mu = p->mu; trk->after_nu = p->nu; trk->after_energy = p->energy;
            trk->interactions_count += 1 + tracker_pop(trk);
            trk->radius = p->r; trk->nu = p->nu; trk->energy = p->energy; trk->shell_id = p->current_shell_id;
            trk->interaction_type = IT_ESCATTERING;
            if ((err = trace_vpacket_volley(c, p, rng, vl, cn))) return err;
        } else
            trk->boundary_buffer += 1;
    }
    trk->boundary_buffer += 1; /* final track_boundary_event (:247-251) */
    return 0;
}

/* ------------------------------------------------------------------------------------- main loop ---- */
int oracle_mc_run(const TardisMcPackets *pk, const TardisMcGeometry *geo, const TardisMcOpacity *op,
                  const TardisMcConfig *cfg, TardisMcResult *res, int math_mode, int n_threads)
{ /* montecarlo_transport.py:238-373 */
    if (!pk || !geo || !op || !cfg || !res) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    if (op->n_shells != geo->n_shells) return TARDIS_MC_ERR_INVALID_ARGUMENT;
    const int64_t P = pk->n_packets, S = op->n_shells, L = op->n_lines, G = cfg->n_spectrum_grid;
    if (n_threads < 1) n_threads = 1;
    if (g_trace_log || g_vtrace_log) n_threads = 1; /* the analysis logs append with a plain counter: serial runs only */
    double *J = res->j_estimator, *nubar = res->nu_bar_estimator, *jb = res->j_blue_estimator, *ed = res->edotlu_estimator;
    double *hist = res->v_packets_energy_hist;
    int own_J = 0, own_nb = 0, own_jb = 0, own_ed = 0, own_h = 0;
    if (!J) { J = (double *)calloc(S, 8); own_J = 1; } else memset(J, 0, S * 8);
    if (!nubar) { nubar = (double *)calloc(S, 8); own_nb = 1; } else memset(nubar, 0, S * 8);
    if (!jb) { jb = (double *)calloc((size_t)L * S, 8); own_jb = 1; } else memset(jb, 0, (size_t)L * S * 8);
    if (!ed) { ed = (double *)calloc((size_t)L * S, 8); own_ed = 1; } else memset(ed, 0, (size_t)L * S * 8);
    if (!hist) { hist = (double *)calloc(G > 0 ? G : 1, 8); own_h = 1; } else memset(hist, 0, G * 8);

    /* n_threads > 1: per-thread private estimator copies reduced at the end -- the reference's own scheme
     * (montecarlo_transport.py:309-314,356-360) -- while they fit in 16 GiB; otherwise omp atomics. */
    const size_t est_elems = (size_t)2 * S + (size_t)2 * L * S;
    int use_private = n_threads > 1 && (double)est_elems * 8.0 * n_threads <= 16.0 * 1024 * 1024 * 1024;
    double *priv = NULL;
    if (use_private) {
        priv = (double *)calloc(est_elems * (size_t)n_threads, 8);
        if (!priv) use_private = 0;
    }
    run_ctx c;
    c.geo = geo; c.op = op; c.cfg = cfg; c.math_mode = math_mode; c.atomic = (n_threads > 1) && !use_private;
    c.J = J; c.nubar = nubar; c.jblue = jb; c.edot = ed;
    const double delta_nu = G >= 2 ? cfg->spectrum_frequency_grid[1] - cfg->spectrum_frequency_grid[0] : 1.0;
    const int track_v = cfg->enable_vpacket_tracking && cfg->number_of_vpackets > 0;
    vlist *vlists = NULL; /* per-packet lists kept only when the consolidated log is requested */
    if (track_v) vlists = (vlist *)calloc(P > 0 ? P : 1, sizeof(vlist));

    counters total;
    memset(&total, 0, sizeof total);
    int64_t first_err_packet = -1;
    int first_err_code = 0;

#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        counters cn;
        memset(&cn, 0, sizeof cn);
        mt_state rng;
        vlist local;
        memset(&local, 0, sizeof local);
        run_ctx tc = c; /* thread-local view: private estimator block when use_private */
        if (use_private) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            double *b = priv + est_elems * (size_t)tid;
            tc.J = b; tc.nubar = b + S; tc.jblue = b + 2 * S; tc.edot = b + 2 * S + (size_t)L * S;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t i = 0; i < P; ++i) {
            rpacket p;
            p.r = pk->initial_radii[i]; p.mu = pk->initial_mus[i]; p.nu = pk->initial_nus[i];
            p.energy = pk->initial_energies[i];
            p.current_shell_id = 0; p.status = ST_IN_PROCESS; p.next_line_id = 0;
            mt_seed(&rng, (uint32_t)pk->packet_seeds[i]);
            tracker trk;
            tracker_init(&trk);
            vlist *vl = track_v ? &vlists[i] : &local;
            vl->n = 0;
            int err = packet_propagation(&tc, &p, &rng, &trk, vl, &cn);
            cn.c[TARDIS_MC_CNT_RNG_DRAWS] += rng.draws;
            if (err) {
#ifdef _OPENMP
#pragma omp critical
#endif
                if (first_err_packet < 0 || i < first_err_packet) { first_err_packet = i; first_err_code = err; }
                continue;
            }
            /* set_packet_collection_output (:70-90) */
            if (res->output_nus) res->output_nus[i] = p.nu;
            if (res->output_energies) {
                if (p.status == ST_REABSORBED) res->output_energies[i] = -p.energy;
                else if (p.status == ST_EMITTED) res->output_energies[i] = p.energy;
            }
            if (res->li_radius) res->li_radius[i] = trk.radius;
            if (res->li_nu) res->li_nu[i] = trk.nu;
            if (res->li_energy) res->li_energy[i] = trk.energy;
            if (res->li_before_nu) res->li_before_nu[i] = trk.before_nu;
            if (res->li_before_mu) res->li_before_mu[i] = trk.before_mu;
            if (res->li_before_energy) res->li_before_energy[i] = trk.before_energy;
            if (res->li_after_nu) res->li_after_nu[i] = trk.after_nu;
            if (res->li_after_mu) res->li_after_mu[i] = trk.after_mu;
            if (res->li_after_energy) res->li_after_energy[i] = trk.after_energy;
            if (res->li_shell_id) res->li_shell_id[i] = trk.shell_id;
            if (res->li_interaction_type) res->li_interaction_type[i] = trk.interaction_type;
            if (res->li_line_absorb_id) res->li_line_absorb_id[i] = trk.line_absorb_id;
            if (res->li_line_emit_id) res->li_line_emit_id[i] = trk.line_emit_id;
            if (res->li_interactions_count) res->li_interactions_count[i] = trk.interactions_count;
            /* add_vpacket_collection_to_histogram (:166-195) */
            for (int64_t j = 0; j < vl->n; ++j) {
                double nu = vl->nus[j];
                if (nu < cfg->spectrum_frequency_grid[0] || nu > cfg->spectrum_frequency_grid[G - 1]) continue;
                int64_t idx = (int64_t)floor((nu - cfg->spectrum_frequency_grid[0]) / delta_nu);
                if (n_threads > 1) {
#pragma omp atomic
                    hist[idx] += vl->energies[j];
                } else
                    hist[idx] += vl->energies[j];
            }
        }
        free(local.nus); free(local.energies); free(local.mus); free(local.rs);
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int k = 0; k < TARDIS_MC_N_COUNTERS; ++k) total.c[k] += cn.c[k];
    }
    if (use_private) { /* estimators.increment(thread copy) for every thread */
        for (int t = 0; t < n_threads; ++t) {
            const double *b = priv + est_elems * (size_t)t;
            for (int64_t k = 0; k < S; ++k) { J[k] += b[k]; nubar[k] += b[S + k]; }
            const double *bj = b + 2 * S, *be = b + 2 * S + (size_t)L * S;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads)
#endif
            for (int64_t k = 0; k < L * S; ++k) { jb[k] += bj[k]; ed[k] += be[k]; }
        }
        free(priv);
    }
    total.c[TARDIS_MC_CNT_PACKETS] = P;
    memcpy(res->counters, total.c, sizeof total.c);
    res->first_error_packet = first_err_packet;
    res->error_code = first_err_code;

    res->vpacket_log_count = 0;
    if (track_v) { /* consolidate_vpacket_tracker (packet_collections.py:310-396) */
        int64_t n = 0;
        for (int64_t i = 0; i < P; ++i) {
            for (int64_t j = 0; j < vlists[i].n; ++j, ++n) {
                if (n < res->vpacket_log_capacity) {
                    if (res->vpacket_nus) res->vpacket_nus[n] = vlists[i].nus[j];
                    if (res->vpacket_energies) res->vpacket_energies[n] = vlists[i].energies[j];
                    if (res->vpacket_initial_mus) res->vpacket_initial_mus[n] = vlists[i].mus[j];
                    if (res->vpacket_initial_rs) res->vpacket_initial_rs[n] = vlists[i].rs[j];
                }
            }
            free(vlists[i].nus); free(vlists[i].energies); free(vlists[i].mus); free(vlists[i].rs);
        }
        res->vpacket_log_count = n;
        free(vlists);
    }
    if (own_J) free(J);
    if (own_nb) free(nubar);
    if (own_jb) free(jb);
    if (own_ed) free(ed);
    if (own_h) free(hist);
    return first_err_code;
}

/* ------------------------------------------------------------- leaf entry points for known-answer tests */
void oracle_mt19937_random(uint32_t seed, int64_t n, double *out)
{
    mt_state s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_random(&s);
}
double oracle_log(double x, int math_mode) { return math_mode ? pm_log(x) : log(x); }
double oracle_exp(double x, int math_mode) { return math_mode ? pm_exp(x) : exp(x); }
void oracle_log_array(const double *x, double *y, int64_t n, int math_mode)
{ for (int64_t i = 0; i < n; ++i) y[i] = math_mode ? pm_log(x[i]) : log(x[i]); }
void oracle_exp_array(const double *x, double *y, int64_t n, int math_mode)
{ for (int64_t i = 0; i < n; ++i) y[i] = math_mode ? pm_exp(x[i]) : exp(x[i]); }
double oracle_doppler_factor(double v, double mu, int full) { return doppler_factor(v, mu, full); }
double oracle_inverse_doppler_factor(double v, double mu, int full) { return inverse_doppler_factor(v, mu, full); }
double oracle_angle_aberration_cmf_to_lf(double r, double t, double mu) { return aberration_cmf_to_lf(r, t, mu); }
double oracle_angle_aberration_lf_to_cmf(double r, double t, double mu) { return aberration_lf_to_cmf(r, t, mu); }
void oracle_distance_boundary(double r, double mu, double r_inner, double r_outer, double *d, int64_t *delta)
{ distance_boundary(r, mu, r_inner, r_outer, d, delta); }
int oracle_distance_line(double nu, double r, double mu, double comov_nu, int is_last, double nu_line, double t,
                         int full, double *d)
{ return distance_line(nu, r, mu, comov_nu, is_last, nu_line, t, full, d); }

/* One call of a single leaf on a caller-described packet.  pkt = {r, mu, nu, energy}, ids = {next_line_id,
 * current_shell_id, status}.  what: 0 trace_packet, 1 move_r_packet(distance=arg), 2 thomson_scatter,
 * 3 line_scatter_event, 4 move_packet_across_shell_boundary(delta=(int)arg), 5 trace_vpacket_volley.
 * The RNG is seeded with `seed` first.  out = {distance, interaction_type, delta_shell}. */
int oracle_packet_step(int what, double *pkt, int64_t *ids, uint32_t seed, double arg, const TardisMcGeometry *geo,
                       const TardisMcOpacity *op, const TardisMcConfig *cfg, TardisMcResult *res, int math_mode,
                       double *out)
{
    run_ctx c;
    c.geo = geo; c.op = op; c.cfg = cfg; c.math_mode = math_mode; c.atomic = 0;
    c.J = res->j_estimator; c.nubar = res->nu_bar_estimator; c.jblue = res->j_blue_estimator; c.edot = res->edotlu_estimator;
    rpacket p;
    p.r = pkt[0]; p.mu = pkt[1]; p.nu = pkt[2]; p.energy = pkt[3];
    p.next_line_id = ids[0]; p.current_shell_id = ids[1]; p.status = ids[2];
    mt_state rng;
    mt_seed(&rng, seed);
    counters cn;
    memset(&cn, 0, sizeof cn);
    int err = 0;
    const int full = cfg->enable_full_relativity;
    if (what == 0) {
        double chi = arg, d; int type; int64_t delta;
        err = trace_packet(&c, &p, &rng, chi, &d, &type, &delta, &cn);
        out[0] = d; out[1] = type; out[2] = (double)delta;
    } else if (what == 1) move_r_packet(&c, &p, arg, full);
    else if (what == 2) thomson_scatter(&c, &p, &rng);
    else if (what == 3) err = line_scatter_event(&c, &p, &rng, &cn);
    else if (what == 4) move_across_shell_boundary(&p.current_shell_id, &p.status, (int64_t)arg, geo->n_shells);
    else if (what == 5) {
        vlist vl; memset(&vl, 0, sizeof vl);
        err = trace_vpacket_volley(&c, &p, &rng, &vl, &cn);
        int64_t n = vl.n < res->vpacket_log_capacity ? vl.n : res->vpacket_log_capacity;
        for (int64_t j = 0; j < n; ++j) {
            res->vpacket_nus[j] = vl.nus[j]; res->vpacket_energies[j] = vl.energies[j];
            res->vpacket_initial_mus[j] = vl.mus[j]; res->vpacket_initial_rs[j] = vl.rs[j];
        }
        res->vpacket_log_count = vl.n;
        free(vl.nus); free(vl.energies); free(vl.mus); free(vl.rs);
    } else return TARDIS_MC_ERR_INVALID_ARGUMENT;
    pkt[0] = p.r; pkt[1] = p.mu; pkt[2] = p.nu; pkt[3] = p.energy;
    ids[0] = p.next_line_id; ids[1] = p.current_shell_id; ids[2] = p.status;
    memcpy(res->counters, cn.c, sizeof cn.c);
    return err;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
