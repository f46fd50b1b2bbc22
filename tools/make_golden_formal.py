"""DEV-CONTAINER-ONLY: golden vectors of the formal integral from the reference's own implementation.

Imports tardis/spectrum/formal_integral/formal_integral_numba.py UNMODIFIED in pure-Python mode (tools/ref_shim.py) and
runs `numba_formal_integral` on small synthetic problems; inputs and outputs are committed as tests/golden/formal_*.npz.
    NPY_DISABLE_CPU_FEATURES is set like for the transport goldens so that np.exp is glibc's exp.
"""
import os
import sys

if os.environ.get("NPY_DISABLE_CPU_FEATURES") is None:
    os.environ["NPY_DISABLE_CPU_FEATURES"] = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX2 FMA3"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
sys.path.insert(0, ref_shim.REF_ROOT)
from tardis.spectrum.formal_integral import formal_integral_numba as fi  # noqa: E402

from tardis_amd import synthetic  # noqa: E402


class Geo:
    def __init__(self, g):
        self.r_inner, self.r_outer = g.r_inner, g.r_outer


class Plasma:
    def __init__(self, nu):
        self.line_list_nu = nu


def make_case(seed, n_shells, n_lines, n_nu, n_p, t_inner=1.0e4, log_tau_mean=-1.0):
    prob = synthetic.make_problem(seed=seed, n_packets=1, n_shells=n_shells, n_lines=n_lines, log_tau_mean=log_tau_mean)
    rng = np.random.default_rng(1000 + seed)
    nu_lines = prob.opacity_state.line_list_nu
    tau = prob.opacity_state.tau_sobolev                      # [L, S]
    n_e = prob.opacity_state.electron_density
    # plausible magnitudes: dilute black-body mean intensities, S_ul (1 - exp(-tau))
    bb = np.array([fi.intensity_black_body(nu, t_inner) for nu in nu_lines])
    w = 0.5 * (prob.geometry.r_inner[0] / prob.geometry.r_outer) ** 2                      # [S]
    jblue = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (n_shells, n_lines))).ravel()  # shell-major flat
    jred = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (n_shells, n_lines))).ravel()
    att = (bb[None, :] * w[:, None] * rng.uniform(0.2, 1.2, (n_shells, n_lines)) * (1 - np.exp(-tau.T))).ravel()
    lo, hi = nu_lines[-1] * 1.02, nu_lines[0] * 0.98
    freqs = np.sort(rng.uniform(lo, hi, n_nu))
    L, I = fi.numba_formal_integral(Geo(prob.geometry), prob.time_explosion, Plasma(nu_lines), t_inner, freqs, att, jred, jblue,
                                    tau, n_e, n_p)
    return dict(r_inner=prob.geometry.r_inner, r_outer=prob.geometry.r_outer, time_explosion=prob.time_explosion,
                line_list_nu=nu_lines, tau_sobolev=tau, electron_density=n_e, inner_temperature=t_inner, frequencies=freqs,
                att_S_ul=att, Jred_lu=jred, Jblue_lu=jblue, n_impact_parameters=n_p, luminosity_densities=L, intensities_nu_p=I)


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    cases = {"formal_small": make_case(1, 5, 400, 24, 17), "formal_thick": make_case(2, 8, 1500, 16, 33, log_tau_mean=0.5),
             "formal_one_shell": make_case(3, 1, 200, 12, 9)}
    for name, c in cases.items():
        np.savez_compressed(os.path.join(out, name + ".npz"), **c)
        print(name, c["luminosity_densities"][:3], np.isfinite(c["luminosity_densities"]).all())
