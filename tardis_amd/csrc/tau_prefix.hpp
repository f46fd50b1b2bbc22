// tau_prefix.hpp -- prefix sums of the Sobolev optical depths along the line list, per shell: what decides a v-packet's
// Russian roulette without walking its lines.
//
// trace_vpacket (tardis/transport/montecarlo/packets/virtual_packet.py:179-244) adds, shell after shell, the electron-scattering
// depth to the boundary and the Sobolev depth of every line the v-packet's comoving frequency passes (trace_vpacket_within_shell,
// :82-175: a serial sum in line order), and after every shell tests `tau > tau_russian` (10): with the default survival
// probability 0 the v-packet is then dropped with energy 0 -- whatever its optical depth was.  On optically thick ejecta that
// is the fate of almost every v-packet (99.96 % on the BASELINE configs[4] table shape), i.e. the serial sums are computed to
// answer a yes/no question.  The answer does not need the serial sum: with P_s[i] = sum_{j<i} tau[s][j],
//     tau_shell = chi d_boundary + (P_s[e] - P_s[start])            (e: first line at or beyond the boundary, from the index tables)
// differs from the reference's serially rounded value by at most a rigorous margin m (below), so
//     tau - m > tau_russian   =>  the reference's test is true,        tau + m < tau_russian   =>  it is false,
// and only a v-packet whose total comes within m of the threshold at some boundary, or that leaves the grid alive (its energy
// needs exp(-tau) of the reference's own sum), is traced line by line (vp_trace / vp_shell_step, unchanged).  Two 8-byte reads per
// shell crossing instead of ~40 optical depths.
//
// P is accumulated in double-double (two-sum; -ffp-contract=off) and rounded once: |P^ - P| <= 2^-53 P.  Margin of one crossing
// (n lines, all tau >= 0 -- a negative optical depth anywhere switches the screening off):
//     |serial - exact| <= 1.01 n 2^-53 (chi d + seg)        (standard bound of recursive summation)
//     |(P^[e] - P^[start]) - seg| <= 2^-53 (P[e] + P[start]) + 2^-53 seg <= 2^-52 rowsum + 2^-53 seg
// plus one rounding per shell of the running total on either side; the kernels use 2.3e-16 (rowsum + (n + 4)(chi d + seg) + 2 tau)
// per crossing and double the sum.
#pragma once
#include <hip/hip_runtime.h>

namespace mc {

struct DD { double hi, lo; };
__device__ __forceinline__ void dd_add(DD &a, double x)
{
    const double s = a.hi + x;
    const double bb = s - a.hi;
    const double err = (a.hi - (s - bb)) + (x - bb);
    a.hi = s;
    a.lo += err;
}
__device__ __forceinline__ void dd_add_dd(DD &a, const DD &b)
{
    dd_add(a, b.hi);
    a.lo += b.lo;
}

// one 256-thread workgroup per shell: pfx[s][0..L] (row stride L + 1), rowsum[s] = pfx[s][L]; *negative |= any tau < 0 (or NaN)
__global__ void __launch_bounds__(256) tau_prefix_kernel(const double *__restrict__ tau_t, int n_lines, double *__restrict__ pfx,
                                                         double *__restrict__ rowsum, int *__restrict__ negative)
{
    __shared__ DD part[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const double *__restrict__ row = tau_t + (size_t)s * (size_t)n_lines;
    double *__restrict__ out = pfx + (size_t)s * (size_t)(n_lines + 1);
    const int per = (n_lines + 255) / 256;
    const int i0 = min(tid * per, n_lines), i1 = min(i0 + per, n_lines);
    DD acc = {0.0, 0.0};
    bool neg = false;
    for (int i = i0; i < i1; ++i) {
        const double x = row[i];
        neg |= !(x >= 0.0);
        dd_add(acc, x);
    }
    part[tid] = acc;
    __syncthreads();
    if (tid == 0) {  // exclusive scan of the 256 chunk totals (serial: 256 double-double adds)
        DD run = {0.0, 0.0};
        for (int k = 0; k < 256; ++k) {
            const DD t = part[k];
            part[k] = run;
            dd_add_dd(run, t);
        }
        rowsum[s] = run.hi + run.lo;
    }
    __syncthreads();
    acc = part[tid];
    for (int i = i0; i < i1; ++i) {
        out[i] = acc.hi + acc.lo;
        dd_add(acc, row[i]);
    }
    if (i1 == n_lines && i0 < n_lines) out[n_lines] = acc.hi + acc.lo;
    if (n_lines == 0 && tid == 0) out[0] = 0.0;
    if (neg) atomicOr(negative, 1);
}

}  // namespace mc
