/* TEST INFRASTRUCTURE (oracle): plain-C restatement of the reference's formal integral.
 *
 * Follows numba_formal_integral and its helpers (tardis/spectrum/formal_integral/formal_integral_numba.py:19-118 intersection
 * points, :120-166 line search, :179-262 initialisation, :264-322 electron-scattering optical depth, :375-560 the ray loop)
 * and tardis/spectrum/formal_integral/base.py:13-16,105-120 (constants, impact parameters, black body).  Pinned against
 * golden vectors produced by the reference itself (tools/make_golden_formal.py, tests/golden/formal_*.npz).
 * Not part of the product: only tests/ and bench-side checks load it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define C_INV 3.33564e-11
#define KB_CGS 1.3806488e-16
#define H_CGS 6.62606957e-27
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define SIGMA_THOMSON 6.652458734e-25 /* tardis/constants.py:1 (astropy const13) sigma_T in cm^2 */

static double intensity_black_body(double nu, double t)
{
    if (nu == 0) return NAN;
    double beta_rad = 1 / (KB_CGS * t);
    double coefficient = 2 * H_CGS * C_INV * C_INV;
    return coefficient * nu * nu * nu / (exp(H_CGS * nu * beta_rad) - 1);
}

static double calculate_intersection_point(double radius, double p, double inv_t)
{
    if (radius > p) return sqrt(radius * radius - p * p) * C_INV * inv_t;
    return 0;
}

/* formal_integral_numba.py:52-118 */
int oracle_fi_intersection_points(int n, const double *r_inner, const double *r_outer, double time_explosion, double p,
                                  double *z, long long *shell_ids)
{
    double inv_t = 1 / time_explosion;
    int offset = n;
    if (p <= r_inner[0]) {
        for (int i = 0; i < n; ++i) {
            z[i] = 1 - calculate_intersection_point(r_outer[i], p, inv_t);
            shell_ids[i] = i;
        }
        return n;
    }
    for (int i = 0; i < n; ++i) {
        double ip = calculate_intersection_point(r_outer[i], p, inv_t);
        if (ip == 0) continue;
        if (offset == n) offset = i;
        int i_low = n - i - 1, i_up = n + i - 2 * offset;
        z[i_low] = 1 + ip; shell_ids[i_low] = i;
        z[i_up] = 1 - ip; shell_ids[i_up] = i;
    }
    return 2 * (n - offset);
}

/* number of lines with nu_line > x (line_search / reverse_binary_search / the searchsorted of the ray loop all reduce to it) */
static int count_greater(const double *nu_line, int n, double x)
{
    int lo = 0, hi = n; /* first index with nu_line[i] <= x */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (nu_line[mid] > x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

int oracle_formal_integral(int n_shells, const double *r_inner, const double *r_outer, double time_explosion, int n_lines,
                           const double *line_list_nu, const double *tau_sobolev /* [L][S] */, const double *electron_density,
                           double inner_temperature, int n_nu, const double *frequencies, const double *att_S_ul,
                           const double *Jred_lu, const double *Jblue_lu, int N, double *luminosity_densities,
                           double *intensities_nu_p /* [n_nu][N] */)
{
    const long long total = (long long)n_shells * n_lines;
    double *exp_tau = malloc(sizeof(double) * (size_t)total);
    for (int s = 0; s < n_shells; ++s)
        for (int l = 0; l < n_lines; ++l) exp_tau[(long long)s * n_lines + l] = exp(-tau_sobolev[(long long)l * n_shells + s]);
    const double radius_max = r_outer[n_shells - 1];
    double *z = malloc(sizeof(double) * 2 * n_shells);
    long long *sid = malloc(sizeof(long long) * 2 * n_shells);
    memset(intensities_nu_p, 0, sizeof(double) * (size_t)n_nu * N);
    for (int nu_idx = 0; nu_idx < n_nu; ++nu_idx) {
        double *I_nu = intensities_nu_p + (size_t)nu_idx * N;
        const double nu = frequencies[nu_idx];
        for (int p_idx = 1; p_idx < N; ++p_idx) {
            /* np.arange(N) * radius_max / (N - 1) */
            const double p = (double)p_idx * radius_max / (double)(N - 1);
            const int n_int = oracle_fi_intersection_points(n_shells, r_inner, r_outer, time_explosion, p, z, sid);
            double I = (p <= r_inner[0]) ? intensity_black_body(nu * z[0], inner_temperature) : 0.0;
            const double nu_start = nu * z[0];
            double intersection_start = time_explosion / C_INV * (1.0 - z[0]);
            long long line_idx = count_greater(line_list_nu, n_lines, nu_start);
            long long off = line_idx + sid[0] * n_lines, jred = off;
            int first = 1;
            double escat = 0;
            for (int i = 0; i < n_int - 1; ++i) {
                const double escat_opacity = electron_density[sid[i]] * SIGMA_THOMSON;
                const double nu_end = nu * z[i + 1];
                const long long nu_end_idx = count_greater(line_list_nu, n_lines, nu_end);
                for (long long k = line_idx; k < nu_end_idx; ++k) {
                    const double intersection_end = time_explosion / C_INV * (1.0 - line_list_nu[line_idx] / nu);
                    const double jb = (off >= 0 && off < total) ? Jblue_lu[off] : 0.0;
                    const double jr = (jred >= 0 && jred < total) ? Jred_lu[jred] : 0.0;
                    if (first == 1) {
                        escat += (intersection_end - intersection_start) * escat_opacity * (jb - I);
                        first = 0;
                    } else {
                        const double avg = 0.5 * (jr + jb);
                        escat += (intersection_end - intersection_start) * escat_opacity * (avg - I);
                        jred += 1;
                    }
                    I += escat;
                    I *= exp_tau[off];
                    I += att_S_ul[off];
                    escat = 0;
                    intersection_start = intersection_end;
                    line_idx += 1;
                    off += 1;
                }
                {
                    const double jb = (off >= 0 && off < total) ? Jblue_lu[off] : 0.0;
                    const double jr = (jred >= 0 && jred < total) ? Jred_lu[jred] : 0.0;
                    const double avg = 0.5 * (jr + jb);
                    const double intersection_end = time_explosion / C_INV * (1.0 - nu_end / nu);
                    escat += (intersection_end - intersection_start) * escat_opacity * (avg - I);
                    intersection_start = intersection_end;
                }
                const long long direction = (sid[i + 1] - sid[i]) * n_lines;
                off += direction;
                jred += direction;
            }
            I_nu[p_idx] = I * p;
        }
        /* 8 pi^2 np.trapezoid(I_nu, dx = radius_max / N) */
        const double dx = radius_max / (double)N;
        double sum = 0;
        for (int k = 0; k + 1 < N; ++k) sum += dx * (I_nu[k + 1] + I_nu[k]) / 2.0;
        luminosity_densities[nu_idx] = 8 * M_PI * M_PI * sum;
    }
    free(exp_tau); free(z); free(sid);
    return 0;
}
