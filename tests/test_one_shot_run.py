"""`tardis_mc_run` -- the one-shot form of the boundary (include/tardis_mc.h; what a ctypes binding inside
MCTransportSolverClassic.run_classic binds when nothing is to stay resident, modes/classic/solver.py:223-234) -- called itself,
held to the reference-generated goldens and the oracle: without v-packets, with the v-packet log sized by the library, with a
caller-sized log (the capacity is the caller's for THAT call only), and with a log that is too small (the count says so; the
entries that fit are the first ones of the consolidated log).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11


def _oracle(oracle, prob):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE)


def _run(eng, prob, **kw):
    return eng.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                   prob.spectrum_frequency_grid, **kw)


def _check(res, ref, g):
    assert np.array_equal(res.output_nus, ref.output_nus) and np.array_equal(res.output_energies, ref.output_energies)
    assert_allclose(res.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(res.output_energies, g["output_energies"], rtol=1e-13, atol=0)
    for f in _golden.TRACKER_I64:
        assert np.array_equal(getattr(res.trackers, f), g["trk_" + f]), f
    for f in _golden.TRACKER_F64:
        assert np.array_equal(getattr(res.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(res.j_estimator, ref.j_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(res.nu_bar_estimator, ref.nu_bar_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(res.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(res.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(res.v_packets_energy_hist, g["v_packets_energy_hist"], rtol=EST_RTOL, atol=0)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws", "packets"):
        assert res.counters[k] == ref.counters[k], k


@pytest.mark.parametrize("name", ["macroatom_nv0", "downbranch_nv0", "scatter_single_shell", "downbranch_fullrel"])
def test_one_shot_run_without_vpackets(oracle, name):
    from tardis_amd.engine import Engine
    prob, g = _golden.load_case(name)
    ref = _oracle(oracle, prob)
    with Engine(0) as eng:
        _check(_run(eng, prob), ref, g)
        _check(_run(eng, prob), ref, g)  # the context is reusable: a second one-shot call re-uploads everything


def _same_log(res, ref, n):  # (get_results consolidates the log in the reference's order: by packet, then by v-packet, packet_collections.py:310-396)
    assert np.array_equal(res.vpacket_nus[:n], ref.vpacket_nus) and np.array_equal(res.vpacket_energies[:n], ref.vpacket_energies)
    assert np.array_equal(res.vpacket_initial_mus[:n], ref.vpacket_initial_mus) and np.array_equal(res.vpacket_initial_rs[:n], ref.vpacket_initial_rs)


def test_one_shot_run_with_the_vpacket_log(oracle):
    from tardis_amd.engine import Engine
    prob, g = _golden.load_case("macroatom_nv3_log")
    ref = _oracle(oracle, prob)
    n_log = len(ref.vpacket_nus)
    assert n_log > 0 and n_log == len(g["vpacket_nus"])
    with Engine(0) as eng:
        for cap in (None, n_log, n_log + 7):  # sized by the library; exactly the caller's; a little more than needed
            res = _run(eng, prob, vpacket_log_capacity=cap)
            _check(res, ref, g)
            assert res.vpacket_log_count == n_log
            _same_log(res, ref, n_log)
            assert_allclose(res.vpacket_nus[:n_log], g["vpacket_nus"], rtol=1e-13, atol=0)
        # a caller's log that is too small: the call succeeds, the count reports what the run produced, and the capacity was the
        # caller's for that call only -- the next call (library-sized) holds everything again
        small = max(n_log // 3, 1)
        res = _run(eng, prob, vpacket_log_capacity=small)
        assert res.vpacket_log_count == n_log
        assert np.array_equal(res.output_nus, ref.output_nus)
        assert_allclose(res.v_packets_energy_hist, g["v_packets_energy_hist"], rtol=EST_RTOL, atol=0)
        assert np.all(np.isin(res.vpacket_nus[:small], ref.vpacket_nus))
        res = _run(eng, prob)
        assert res.vpacket_log_count == n_log
        _same_log(res, ref, n_log)


def test_one_shot_run_reports_the_reference_errors(oracle):
    """An unsorted line list: the reference raises MonteCarloException at the first packet that meets it; so does the one-shot call,
    with the failing packet's index attached."""
    from tardis_amd.engine import Engine, MonteCarloException
    prob, g = _golden.load_case("downbranch_nv0")
    op = prob.opacity_state
    nu = op.line_list_nu.copy()
    nu[len(nu) // 2:] = nu[len(nu) // 2:][::-1]
    bad = st.OpacityState(op.electron_density, op.t_electrons, nu, op.tau_sobolev, op.transition_probabilities, op.line2macro_level_upper,
                          op.macro_block_edge_index, op.transition_type, op.destination_level_id, op.transition_line_id)
    with Engine(0) as eng:
        with pytest.raises(MonteCarloException) as ei:
            eng.run(prob.packet_collection, prob.geometry, prob.time_explosion, bad, prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        assert ei.value.packet_index >= 0


@pytest.mark.gpu
def test_the_one_shot_call_streams_its_results(oracle):
    """tardis_mc_run knows the result arrays from the start: a call of several launches fills them launch by launch (tardis_mc_stream_results) --
    same bits as the oracle's."""
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=41, n_packets=700_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                     prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())
    with Engine(0) as eng:
        eng.set_option("log_capacity", 1 << 21); eng.set_option("stream_min_packets", 4096)
        got = eng.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid)
        assert eng.streamed_packets()[0] > 0 and eng.last_kernel_times()["launches"] >= 3
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    for f in ("shell_id", "interaction_type", "interaction_line_absorb_id", "interaction_line_emit_id", "interactions_count"):
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
