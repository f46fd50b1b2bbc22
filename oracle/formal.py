"""TEST INFRASTRUCTURE (oracle): ctypes front-end of oracle/formal_integral.c (the reference's formal integral, restated)."""
import ctypes as C

import numpy as np

from . import oracle as _o


def formal_integral(r_inner, r_outer, time_explosion, line_list_nu, tau_sobolev, electron_density, inner_temperature,
                    frequencies, att_S_ul, Jred_lu, Jblue_lu, n_impact_parameters):
    """numba_formal_integral (tardis/spectrum/formal_integral/formal_integral_numba.py:375-560) on plain arrays:
    returns (luminosity_densities[n_nu], intensities_nu_p[n_nu, N])."""
    lib = _o.lib()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    r_inner, r_outer, nu_l, tau, n_e = f(r_inner), f(r_outer), f(line_list_nu), f(tau_sobolev), f(electron_density)
    freqs, att, jred, jblue = f(frequencies), f(att_S_ul), f(Jred_lu), f(Jblue_lu)
    S, Ln, n_nu, N = len(r_inner), len(nu_l), len(freqs), int(n_impact_parameters)
    assert tau.shape == (Ln, S) and att.size == S * Ln and jred.size == S * Ln and jblue.size == S * Ln
    L = np.zeros(n_nu)
    I = np.zeros((n_nu, N))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    fn = lib.oracle_formal_integral
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = fn(S, p(r_inner), p(r_outer), float(time_explosion), Ln, p(nu_l), p(tau), p(n_e), float(inner_temperature), n_nu,
            p(freqs), p(att), p(jred), p(jblue), N, p(L), p(I))
    assert rc == 0
    return L, I
