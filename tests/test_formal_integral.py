"""Formal integral: the oracle's C restatement against golden vectors produced by the reference's own
numba_formal_integral (tools/make_golden_formal.py), and -- on the GPU -- the HIP kernels against the oracle and the goldens."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["formal_small", "formal_thick", "formal_one_shell"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    from oracle import formal
    g = load(name)
    L, I = formal.formal_integral(g["r_inner"], g["r_outer"], float(g["time_explosion"]), g["line_list_nu"], g["tau_sobolev"],
                                  g["electron_density"], float(g["inner_temperature"]), g["frequencies"], g["att_S_ul"],
                                  g["Jred_lu"], g["Jblue_lu"], int(g["n_impact_parameters"]))
    # same libm, same operation order per ray: the intensities agree to the last bits; the trapezoid sum differs from
    # numpy's pairwise summation by rounding only
    assert_allclose(I, g["intensities_nu_p"], rtol=1e-13, atol=0)
    assert_allclose(L, g["luminosity_densities"], rtol=1e-13, atol=0)
    assert (g["intensities_nu_p"][:, 1:-1] > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_oracle_and_reference(name):
    """Device formal integral vs the reference's own output.  Tolerance 1e-11: the device evaluates exp() with its own
    correctly rounded routine (the reference with libm / numpy), and a ray is a product of up to thousands of such factors."""
    from oracle import formal
    from tardis_amd.formal_integral import FormalIntegratorHIP

    class Geo:
        pass

    class Plasma:
        pass

    g = load(name)
    geo, pl = Geo(), Plasma()
    geo.r_inner, geo.r_outer = g["r_inner"], g["r_outer"]
    geo.v_inner = g["r_inner"] / float(g["time_explosion"]); geo.v_outer = g["r_outer"] / float(g["time_explosion"])
    pl.line_list_nu = g["line_list_nu"]
    fi = FormalIntegratorHIP(geo, float(g["time_explosion"]), pl, int(g["n_impact_parameters"]))
    L, I = fi.formal_integral(float(g["inner_temperature"]), g["frequencies"], g["att_S_ul"], g["Jred_lu"], g["Jblue_lu"],
                              g["tau_sobolev"], g["electron_density"], int(g["n_impact_parameters"]))
    fi.close()
    assert_allclose(I, g["intensities_nu_p"], rtol=1e-11, atol=0)
    assert_allclose(L, g["luminosity_densities"], rtol=1e-11, atol=0)
    Lo, Io = formal.formal_integral(g["r_inner"], g["r_outer"], float(g["time_explosion"]), g["line_list_nu"], g["tau_sobolev"],
                                    g["electron_density"], float(g["inner_temperature"]), g["frequencies"], g["att_S_ul"],
                                    g["Jred_lu"], g["Jblue_lu"], int(g["n_impact_parameters"]))
    assert_allclose(I, Io, rtol=1e-11, atol=0)
    assert_allclose(L, Lo, rtol=1e-11, atol=0)


@pytest.mark.gpu
def test_hip_formal_integral_tardis_example_shape():
    """The default integrated-spectrum shape of tardis_example (20 shells, 3e4 lines) at a reduced frequency count, against
    the oracle; exercises rays that sweep thousands of lines."""
    from oracle import formal
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=4, n_packets=1, n_shells=20, n_lines=30_000, log_tau_mean=-2.0)
    rng = np.random.default_rng(5)
    S, Ln = 20, 30_000
    nu_l = prob.opacity_state.line_list_nu
    bb = 2 * 6.62606957e-27 * 3.33564e-11**2 * nu_l**3 / np.expm1(6.62606957e-27 * nu_l / (1.3806488e-16 * 1e4))
    w = 0.5 * (prob.geometry.r_inner[0] / prob.geometry.r_outer) ** 2
    jblue = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (S, Ln))).ravel()
    jred = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (S, Ln))).ravel()
    att = (bb[None, :] * w[:, None] * rng.uniform(0.2, 1.2, (S, Ln)) * (1 - np.exp(-prob.opacity_state.tau_sobolev.T))).ravel()
    freqs = np.linspace(nu_l[-1] * 1.05, nu_l[0] * 0.95, 96)
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    L, I = eng.formal_integral(1.0e4, freqs, att, jred, jblue, 200, want_intensities=True)
    eng.close()
    Lo, Io = formal.formal_integral(prob.geometry.r_inner, prob.geometry.r_outer, prob.time_explosion, nu_l,
                                    prob.opacity_state.tau_sobolev, prob.opacity_state.electron_density, 1.0e4, freqs, att, jred,
                                    jblue, 200)
    assert_allclose(I, Io, rtol=1e-10, atol=0)
    assert_allclose(L, Lo, rtol=1e-10, atol=0)
