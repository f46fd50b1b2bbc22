/* TEST INFRASTRUCTURE (oracle) -- portable, nearly-correctly-rounded log/exp in double-double arithmetic.
 *
 * Why: the reference draws tau_event = -np.log(xi) (homologous_rad_packet_transport.py:84) and weights
 * v-packets by math.exp(-tau) (virtual_packet.py:231,372).  libm's log/exp are not bit-defined across
 * libraries (glibc 2.35 log differs from the correctly rounded value for ~0.09 % of arguments; numpy's
 * AVX-512 log differs from glibc for ~0.35 %), and the GPU has no glibc.  To make "HIP == oracle" a
 * BIT-EXACT statement, both sides evaluate the same algorithm below (only +,-,*,fma and table look-ups,
 * compiled with -ffp-contract=off), whose result is the correctly rounded log/exp except for ~1e-4 of
 * arguments.  The oracle can also run with libm (math_mode 0) -- that mode is what is pinned against the
 * reference-generated golden vectors; tests quantify libm-vs-portable differences (<= 1 ulp per call).
 *
 * The HIP copy of this algorithm lives in tardis_amd/csrc/mc_math.hpp (separate file on purpose).
 */
#ifndef ORACLE_PORTABLE_MATH_H
#define ORACLE_PORTABLE_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "portable_math_tables.h"

static const double pm_log_tab[MC_LOG_TABLE_LEN][4] = {MC_LOG_TABLE_VALUES};
static const double pm_log1p_tail[10] = {MC_LOG1P_TAIL_COEFS};
static const double pm_exp_tab[MC_EXP_TABLE_LEN][2] = {MC_EXP_TABLE_VALUES};
static const double pm_exp_tail[8] = {MC_EXP_TAIL_COEFS};

static inline uint64_t pm_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double pm_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* Knuth two-sum: s + e == a + b exactly */
#define PM_TWO_SUM(a, b, s, e) do { double _s = (a) + (b); double _bb = _s - (a); \
    (e) = ((a) - (_s - _bb)) + ((b) - _bb); (s) = _s; } while (0)

/* Domain: x finite, x >= 2^-1022 (normal).  x == 0 -> -inf.  (xi in [0,1) only needs (0,1).) */
static inline double pm_log(double x)
{
    if (x == 0.0) return -INFINITY;
    uint64_t ix = pm_bits(x);
    int64_t e = (int64_t)(ix >> 52) - 1023;
    uint64_t mant = ix & 0x000fffffffffffffULL;
    if (mant > 0x6a09e667f3bcdULL) { e += 1; ix = mant | (0x3feULL << 52); } /* m in [sqrt2/2, 1) */
    else ix = mant | (0x3ffULL << 52);                                       /* m in [1, sqrt2]  */
    double m = pm_from_bits(ix);
    int j = (int)(m * 64.0 + 0.5);
    const double *t = pm_log_tab[j - 45];
    double u = m - (double)j * 0.015625;          /* exact (Sterbenz) */
    /* q = u / c_j in double-double */
    double qh = u * t[0];
    double ql = fma(u, t[0], -qh) + u * t[1];
    /* s = -q^2/2 in double-double */
    double sq = qh * qh;
    double sqe = fma(qh, qh, -sq) + 2.0 * qh * ql;
    double sh = -0.5 * sq, sl = -0.5 * sqe;
    /* tail q^3 * P(q) */
    double p = pm_log1p_tail[9];
    for (int k = 8; k >= 0; --k) p = p * qh + pm_log1p_tail[k];
    double p3 = qh * qh * qh * p;
    /* a = log1p(q) = qh + sh + (ql + sl + p3); |qh| >= |sh| */
    double ah = qh + sh;
    double al = (qh - ah) + sh;
    al = al + (ql + sl + p3);
    /* total = e*ln2 + ln c_j + a */
    double ed = (double)e;
    double kh = ed * MC_LN2_HI;                   /* exact: MC_LN2_HI has 42 significant bits */
    double kl = ed * MC_LN2_LO;
    double s1, e1, s2, e2;
    PM_TWO_SUM(kh, t[2], s1, e1);
    PM_TWO_SUM(s1, ah, s2, e2);
    double low = e1 + e2 + kl + t[3] + al;
    return s2 + low;
}

/* exp(x) for finite x; overflow -> +inf, deep underflow -> 0 */
static inline double pm_exp(double x)
{
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    double kd = x * MC_EXP_INV_LN2_64;
    kd = floor(kd + 0.5);
    int64_t k = (int64_t)kd;
    /* r = x - k*ln2/64 in double-double */
    double rh = x - kd * MC_EXP_LN2_64_HI;        /* product exact (32-bit hi part); difference exact */
    double t1 = kd * MC_EXP_LN2_64_LO;
    double t1e = fma(kd, MC_EXP_LN2_64_LO, -t1);
    double r, re;
    PM_TWO_SUM(rh, -t1, r, re);
    double rl = re - t1e - kd * MC_EXP_LN2_64_LOLO;
    /* p = expm1(r) = r + r^2/2 + r^3*E(r) ; keep (r, pl) */
    double r2 = r * r;
    double r2e = fma(r, r, -r2) + 2.0 * r * rl;
    double ep = pm_exp_tail[7];
    for (int i = 6; i >= 0; --i) ep = ep * r + pm_exp_tail[i];
    double pl = rl + (0.5 * r2 + (0.5 * r2e + r2 * r * ep));
    /* T * (1 + r + pl) */
    int64_t i64 = k & 63;
    int64_t q = (k - i64) / 64;
    double th = pm_exp_tab[i64][0], tl = pm_exp_tab[i64][1];
    double pr = th * r;
    double pre = fma(th, r, -pr);
    double s, se;
    PM_TWO_SUM(th, pr, s, se);
    double low = se + pre + th * pl + tl + tl * r;
    double res = s + low;
    return ldexp(res, (int)q);
}
#endif
