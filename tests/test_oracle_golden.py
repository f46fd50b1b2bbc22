"""The oracle against golden vectors produced by the reference itself (tools/make_golden.py).

This is what makes the oracle trustworthy as the checker of the HIP engine: every per-packet output, every
estimator cell, every TrackerLastInteraction field and the v-packet histogram/log of the reference's
``montecarlo_transport_with_vpackets`` (modes/montecarlo_transport.py:238-373) are reproduced.

Tolerances: rtol 1e-13 on per-packet quantities and 1e-12 on estimators is the reference's own regression
tolerance (tests/test_montecarlo_main_loop.py:14-60, test_transport.py:27).  In libm mode the oracle is
additionally required to be BIT-EXACT whenever this machine's libm reproduces the log/exp probe values recorded
on the generating machine (glibc's results can depend on the CPU's ifunc selection).
"""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden

MATH_MODES = {"libm": 0, "portable": 1}


def _libm_matches_generator(oracle):
    p = np.load(os.path.join(_golden.GOLDEN_DIR, "libm_probe.npz"))
    return (np.array_equal(oracle.log_array(p["x"], 0), p["log_x"])
            and np.array_equal(oracle.exp_array(-30 * p["x"], 0), p["exp_mx"]))


@pytest.mark.parametrize("math_mode", list(MATH_MODES))
@pytest.mark.parametrize("name", _golden.CASES)
def test_oracle_reproduces_reference(oracle, name, math_mode):
    prob, g = _golden.load_case(name)
    r = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state,
                   prob.montecarlo_configuration, prob.spectrum_frequency_grid, math_mode=MATH_MODES[math_mode])
    assert r.return_code == 0 and r.first_error_packet == -1
    stride = int(g["line_estimator_stride"])
    # integer / index results: always exact
    assert np.array_equal(np.sign(r.output_energies), np.sign(g["output_energies"]))
    for f in _golden.TRACKER_I64:
        assert np.array_equal(getattr(r.trackers, f), g["trk_" + f]), f
    # floating point
    assert_allclose(r.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(r.output_energies, g["output_energies"], rtol=1e-13, atol=0)
    for f in _golden.TRACKER_F64:
        assert_allclose(getattr(r.trackers, f), g["trk_" + f], rtol=1e-13, atol=0, equal_nan=True, err_msg=f)
    assert np.all(np.isnan(r.trackers.mu))  # never assigned by the reference (tracker_last_interaction.py:67)
    assert_allclose(r.j_estimator, g["j_estimator"], rtol=1e-12, atol=0)
    assert_allclose(r.nu_bar_estimator, g["nu_bar_estimator"], rtol=1e-12, atol=0)
    assert_allclose(r.j_blue_estimator[::stride], g["j_blue_estimator"], rtol=1e-12, atol=0)
    assert_allclose(r.edotlu_estimator[::stride], g["edotlu_estimator"], rtol=1e-12, atol=0)
    assert_allclose(r.j_blue_estimator.sum(axis=0), g["j_blue_shell_sums"], rtol=1e-12, atol=0)
    assert_allclose(r.edotlu_estimator.sum(axis=0), g["edotlu_shell_sums"], rtol=1e-12, atol=0)
    assert_allclose(r.v_packets_energy_hist, g["v_packets_energy_hist"], rtol=1e-12, atol=0)
    if "vpacket_nus" in g:
        assert r.vpacket_log_count == len(g["vpacket_nus"])
        assert_allclose(r.vpacket_nus, g["vpacket_nus"], rtol=1e-13, atol=0)
        assert_allclose(r.vpacket_energies, g["vpacket_energies"], rtol=1e-12, atol=0)
        assert_allclose(r.vpacket_initial_mus, g["vpacket_initial_mus"], rtol=1e-13, atol=0)
        assert_allclose(r.vpacket_initial_rs, g["vpacket_initial_rs"], rtol=1e-13, atol=0)
    if math_mode == "libm" and _libm_matches_generator(oracle):
        for a, b in ((r.output_nus, g["output_nus"]), (r.output_energies, g["output_energies"]),
                     (r.j_estimator, g["j_estimator"]), (r.nu_bar_estimator, g["nu_bar_estimator"]),
                     (r.j_blue_estimator[::stride], g["j_blue_estimator"]),
                     (r.edotlu_estimator[::stride], g["edotlu_estimator"]),
                     (r.v_packets_energy_hist, g["v_packets_energy_hist"])):
            assert np.array_equal(a, b)


def test_portable_math_is_correctly_rounded_on_probe(oracle):
    """pm_log / pm_exp vs libm: at most 1 ulp apart, and equal for >= 99.5 % of arguments."""
    rng = np.random.default_rng(5)
    x = rng.random(200_000)
    a, b = oracle.log_array(x, 0), oracle.log_array(x, 1)
    assert np.max(np.abs(a - b) / np.spacing(np.abs(a))) <= 1.0
    assert np.count_nonzero(a != b) < 0.005 * x.size
    t = -30 * x
    a, b = oracle.exp_array(t, 0), oracle.exp_array(t, 1)
    assert np.max(np.abs(a - b) / np.spacing(np.abs(a))) <= 1.0
    assert np.count_nonzero(a != b) < 0.005 * x.size
    assert oracle.log_array([0.0], 1)[0] == -np.inf and oracle.log_array([1.0], 1)[0] == 0.0
    assert oracle.exp_array([0.0], 1)[0] == 1.0 and oracle.exp_array([-800.0], 1)[0] == 0.0


def test_oracle_thread_count_invariance(oracle):
    """Packets reseed their own RNG, so results do not depend on the thread count except for the summation
    order of the estimators ('thread-count invariance is not tested' in the reference)."""
    prob, g = _golden.load_case("downbranch_nv0")
    args = (prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state,
            prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    r1 = oracle.run(*args, n_threads=1)
    r4 = oracle.run(*args, n_threads=4)
    assert np.array_equal(r1.output_nus, r4.output_nus) and np.array_equal(r1.output_energies, r4.output_energies)
    assert_allclose(r4.j_estimator, r1.j_estimator, rtol=1e-12)
    assert_allclose(r4.j_blue_estimator, r1.j_blue_estimator, rtol=1e-12)
    assert r1.counters == r4.counters
