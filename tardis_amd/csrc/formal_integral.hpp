// formal_integral.hpp -- the formal integral of the spectrum on the device.
//
// Follows numba_formal_integral (tardis/spectrum/formal_integral/formal_integral_numba.py:375-560; the reference's own
// numba.cuda version is formal_integral_cuda.py:272-489) on the resident geometry / line list / Sobolev optical depths:
// for every frequency nu and impact parameter p a ray is followed through the shells, picking up the line source
// function att_S_ul at every resonance, attenuating with exp(-tau_Sobolev) and adding the electron-scattering term of
// Lucy 1999 Eqs. 26-28; the luminosity density is 8 pi^2 times the trapezoid integral of I(p) p over p.
//
// Mapping: one thread per ray, 64 consecutive frequencies of one impact parameter per wave -- neighbouring frequencies
// walk nearly the same lines of the same shells, so the loads of a wave fall into the same cache lines.  The
// intersection points depend on p only and are computed once (fi_intersections_kernel).  exp(-tau) is tabulated once
// per call in the shell-major layout of tau_t.  fp64 throughout, same operation order per ray as the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_math.hpp"

namespace mc {

constexpr double FI_C_INV = 3.33564e-11;  // spectrum/formal_integral/base.py:13-15
constexpr double FI_KB_CGS = 1.3806488e-16;
constexpr double FI_H_CGS = 6.62606957e-27;

__global__ void fi_exp_tau_kernel(const double *__restrict__ tau_t, long long n, double *__restrict__ exp_tau)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        exp_tau[i] = mcm::exp(-tau_t[i]);
}

__device__ __forceinline__ double fi_intersection_point(double radius, double p, double inv_t)
{
    if (radius > p) return sqrt(radius * radius - p * p) * FI_C_INV * inv_t;
    return 0.0;
}

// populate_intersection_points (formal_integral_numba.py:52-118), one thread per impact parameter
__global__ void fi_intersections_kernel(int S, const double *__restrict__ r_inner, const double *__restrict__ r_outer, double t_exp,
                                        int N, double *__restrict__ z /* [N][2S] */, int *__restrict__ sid /* [N][2S] */,
                                        int *__restrict__ n_int /* [N] */)
{
    const int p_idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (p_idx >= N) return;
    const double radius_max = r_outer[S - 1];
    const double p = (double)p_idx * radius_max / (double)(N - 1);  // calculate_impact_parameters (base.py:105-108)
    const double inv_t = 1 / t_exp;
    double *zp = z + (size_t)p_idx * 2 * S;
    int *sp = sid + (size_t)p_idx * 2 * S;
    if (p <= r_inner[0]) {
        for (int i = 0; i < S; ++i) { zp[i] = 1 - fi_intersection_point(r_outer[i], p, inv_t); sp[i] = i; }
        n_int[p_idx] = S;
        return;
    }
    int offset = S;
    for (int i = 0; i < S; ++i) {
        const double ip = fi_intersection_point(r_outer[i], p, inv_t);
        if (ip == 0) continue;
        if (offset == S) offset = i;
        const int i_low = S - i - 1, i_up = S + i - 2 * offset;
        zp[i_low] = 1 + ip; sp[i_low] = i;
        zp[i_up] = 1 - ip; sp[i_up] = i;
    }
    n_int[p_idx] = 2 * (S - offset);
}

struct FormalIntegralArgs {
    int n_shells, n_lines, n_nu, N;
    double t_exp, inner_temperature, sigma_thomson, radius_max;
    const double *r_inner, *nu_line, *n_e, *exp_tau, *att_S_ul, *Jred_lu, *Jblue_lu, *frequencies;
    const double *z;
    const int *sid, *n_int;
    double *intensities_nu_p;  // [n_nu][N]
    unsigned long long *line_steps;  // work counter: resonances crossed by all rays
};

__device__ __forceinline__ double fi_black_body(double nu, double t)
{
    if (nu == 0) return __builtin_nan("");
    const double beta_rad = 1 / (FI_KB_CGS * t);
    const double coefficient = 2 * FI_H_CGS * FI_C_INV * FI_C_INV;
    return coefficient * nu * nu * nu / (mcm::exp(FI_H_CGS * nu * beta_rad) - 1);
}

// the ray loop (formal_integral_numba.py:441-548)
__global__ void __launch_bounds__(64) fi_rays_kernel(FormalIntegralArgs a)
{
    const int nu_idx = blockIdx.x * 64 + threadIdx.x;
    const int p_idx = blockIdx.y;
    if (nu_idx >= a.n_nu) return;
    double *out = a.intensities_nu_p + (size_t)nu_idx * a.N + p_idx;
    if (p_idx == 0) { *out = 0.0; return; }  // (whole blocks: blockIdx.y == 0)
    const int S = a.n_shells, L = a.n_lines;
    const long long total = (long long)S * L;
    const double nu = a.frequencies[nu_idx];
    const double *__restrict__ z = a.z + (size_t)p_idx * 2 * S;
    const int *__restrict__ sid = a.sid + (size_t)p_idx * 2 * S;
    const int n_int = a.n_int[p_idx];
    const double p = (double)p_idx * a.radius_max / (double)(a.N - 1);  // same expression as fi_intersections_kernel
    const double z0 = z[0];
    double I = (p <= a.r_inner[0]) ? fi_black_body(nu * z0, a.inner_temperature) : 0.0;
    double intersection_start = a.t_exp / FI_C_INV * (1.0 - z0);
    // idx_nu_start = number of lines with nu_line > nu_start (line_search, :132-166)
    long long line_idx;
    {
        const double x = nu * z0;
        int lo = 0, hi = L;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.nu_line[mid] > x) lo = mid + 1; else hi = mid;
        }
        line_idx = lo;
    }
    long long off = line_idx + (long long)sid[0] * L, jred = off;
    bool first = true;
    double escat = 0.0;
    unsigned steps = 0;
    for (int i = 0; i < n_int - 1; ++i) {
        const int shell = sid[i];
        const double escat_opacity = a.n_e[shell] * a.sigma_thomson;
        const double nu_end = nu * z[i + 1];
        // lines with nu_line > nu_end are crossed inside this shell segment
        while (line_idx < L) {
            const double nl = a.nu_line[line_idx];
            if (!(nl > nu_end)) break;
            const double intersection_end = a.t_exp / FI_C_INV * (1.0 - nl / nu);
            const double jb = (off >= 0 && off < total) ? a.Jblue_lu[off] : 0.0;
            if (first) {
                escat += (intersection_end - intersection_start) * escat_opacity * (jb - I);
                first = false;
            } else {
                const double jr = (jred >= 0 && jred < total) ? a.Jred_lu[jred] : 0.0;
                const double avg = 0.5 * (jr + jb);
                escat += (intersection_end - intersection_start) * escat_opacity * (avg - I);
                jred += 1;
            }
            I += escat;
            I *= a.exp_tau[off];   // Lucy 1999, Eq 26
            I += a.att_S_ul[off];
            escat = 0.0;
            intersection_start = intersection_end;
            line_idx += 1;
            off += 1;
            ++steps;
        }
        {   // electron scattering to the cell boundary
            const double jb = (off >= 0 && off < total) ? a.Jblue_lu[off] : 0.0;
            const double jr = (jred >= 0 && jred < total) ? a.Jred_lu[jred] : 0.0;
            const double avg = 0.5 * (jr + jb);
            const double intersection_end = a.t_exp / FI_C_INV * (1.0 - nu_end / nu);
            escat += (intersection_end - intersection_start) * escat_opacity * (avg - I);
            intersection_start = intersection_end;
        }
        const long long direction = (long long)(sid[i + 1] - shell) * L;
        off += direction;
        jred += direction;
    }
    *out = I * p;
    unsigned long long st = steps;
    for (int o = 32; o > 0; o >>= 1) st += __shfl_down(st, o);
    if (threadIdx.x == 0 && a.line_steps) atomicAdd(a.line_steps, st);
}

// 8 pi^2 np.trapezoid(I_nu, dx = radius_max / N), one thread per frequency
__global__ void fi_trapezoid_kernel(const double *__restrict__ intensities_nu_p, int n_nu, int N, double radius_max,
                                    double *__restrict__ luminosity_densities)
{
    const int nu_idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (nu_idx >= n_nu) return;
    const double *I = intensities_nu_p + (size_t)nu_idx * N;
    const double dx = radius_max / (double)N;
    double sum = 0.0;
    for (int k = 0; k + 1 < N; ++k) sum += dx * (I[k + 1] + I[k]) / 2.0;
    const double pi = 3.141592653589793;
    luminosity_densities[nu_idx] = 8 * pi * pi * sum;
}

}  // namespace mc
