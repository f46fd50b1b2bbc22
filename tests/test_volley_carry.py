"""The pooled v-packet volleys with the cut-off and carry-over (engine option `vp_carry_min_active`, tardis_amd/csrc/propagate_wave.hpp:
VpPark / WS_VCARRY): a volley phase ends once every item has been handed to a worker lane and at most that many lanes still trace;
those lanes park their v-packets and go on in the next pass's phase, the owners of unfinished rounds wait with the round uncommitted.
Scheduling only -- the function of the inputs is that of every other kernel: per-packet results bit-exact against the oracle and the
reference-generated fixtures (incl. the consolidated v-packet log, whose ORDER is the reference's commit order), histogram and
estimators to the summation-order tolerance, work counters exact.

Reference behaviour held: packets/virtual_packet.py:82-386 (trace_vpacket_volley), classic/packet_propagation.py:109-118,201-244.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st, synthetic

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11
VP_CASES = [n for n in _golden.CASES if "_nv2" in n or "_nv3" in n or "_nv10" in n]


def _oracle(oracle, prob, **kw):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), **kw)


def _engine(prob, **options):
    from tardis_amd.engine import Engine
    eng = Engine(0)
    for k, v in options.items():
        eng.set_option(k, v)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    return eng


def _compare(got, ref, trackers=True, lines=True):
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    if trackers:
        for f in st.LastInteractionTrackers.I64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
        for f in st.LastInteractionTrackers.F64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL)
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.nu_bar_estimator, ref.nu_bar_estimator, rtol=EST_RTOL)
    if lines:
        assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
        assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws", "packets"):
        assert got.counters[k] == ref.counters[k], k


@pytest.mark.parametrize("cut", [1, 8, 16, 63])
@pytest.mark.parametrize("name", VP_CASES)
def test_carry_over_on_the_vpacket_goldens(oracle, name, cut):
    """Every reference-generated v-packet fixture on the wave kernel's pooled volleys with the cut-off at 1 / 8 / 16 / 63 lanes."""
    from tardis_amd import transport
    from tardis_amd.engine import Engine
    prob, g = _golden.load_case(name)
    ref = _oracle(oracle, prob)
    eng = Engine(0)
    try:
        eng.set_option("variant", 2)
        eng.set_option("vp_carry_min_active", cut)
        pc = prob.packet_collection
        pc.output_nus[:] = -99.0; pc.output_energies[:] = -99.0
        trk = st.LastInteractionTrackers(pc.number_of_packets)
        hist, vt, eb, el = transport.montecarlo_transport_with_vpackets(
            pc, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration, prob.spectrum_frequency_grid,
            trk, prob.montecarlo_configuration.NUMBER_OF_VPACKETS, False, None, engine=eng)
        counters = transport.montecarlo_transport_with_vpackets.last_counters
        # (a survival probability > 0 sends the call to the group kernel: the pooled volleys budget one roulette draw per v-packet)
        assert eng.last_variant() == (1 if prob.montecarlo_configuration.SURVIVAL_PROBABILITY > 0 else 2)
    finally:
        eng.close()
    assert np.array_equal(pc.output_nus, ref.output_nus) and np.array_equal(pc.output_energies, ref.output_energies)
    assert_allclose(pc.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    for f in _golden.TRACKER_I64:
        assert np.array_equal(getattr(trk, f), g["trk_" + f]), f
    for f in _golden.TRACKER_F64:
        assert np.array_equal(getattr(trk, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(hist, ref.v_packets_energy_hist, rtol=EST_RTOL, atol=0)
    assert_allclose(hist, g["v_packets_energy_hist"], rtol=EST_RTOL, atol=0)
    assert_allclose(eb.mean_intensity_total, ref.j_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(el.mean_intensity_blueward, ref.j_blue_estimator, rtol=EST_RTOL, atol=0)
    if "vpacket_nus" in g:  # the consolidated log: same entries in the same order
        assert np.array_equal(vt.nus, ref.vpacket_nus) and np.array_equal(vt.energies, ref.vpacket_energies)
        assert np.array_equal(vt.initial_mus, ref.vpacket_initial_mus) and np.array_equal(vt.initial_rs, ref.vpacket_initial_rs)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws", "packets"):
        assert counters[k] == ref.counters[k], k


@pytest.mark.parametrize("shape", [
    dict(n_shells=20, n_lines=3_000, line_interaction_type="downbranch", n_vpackets=10),
    dict(n_shells=60, n_lines=20_000, line_interaction_type="macroatom", n_vpackets=7),
    dict(n_shells=8, n_lines=1_000, line_interaction_type="scatter", n_vpackets=3, enable_full_relativity=True),
    dict(n_shells=100, n_lines=300_000, line_interaction_type="macroatom", n_vpackets=10, level_sizes="heavy"),
], ids=["downbranch-nv10", "macroatom-60shells-nv7", "scatter-fullrel-nv3", "100shells-screened-nv10"])
@pytest.mark.parametrize("cut", [8, 24])
def test_carry_over_matches_the_oracle(oracle, shape, cut):
    """Enough packets that every wave is full and phases really are cut (the device counter of parked v-packets is not zero
    on the multi-shell shapes); the last shape is screened on prefix sums (n_lines >= 2500 n_shells)."""
    n = 6_000 if shape["n_shells"] >= 100 else 40_000
    prob = synthetic.make_problem(seed=11, n_packets=n, **shape)
    ref = _oracle(oracle, prob)
    eng = _engine(prob, variant=2, vp_carry_min_active=cut)
    try:
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        assert eng.last_variant() == 2
        _compare(got, ref)
        # the same call again (the engine's buffers -- parked v-packets, scratch results -- are reused)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        _compare(eng.get_results(track_last_interaction=True), ref)
    finally:
        eng.close()


def test_carry_over_survives_suspended_epochs(oracle):
    """A call split into many launches by a small line-visit log: waves are suspended with rounds handed over and v-packets parked
    (neither is saved: the owners hand their round over again after the resume)."""
    prob = synthetic.make_problem(seed=31, n_packets=140_001, n_shells=12, n_lines=2_000, line_interaction_type="downbranch", n_vpackets=5)
    ref = _oracle(oracle, prob, track_last_interaction=False)
    eng = _engine(prob, variant=2, vp_carry_min_active=16, log_capacity=1 << 19, track_last_interaction=0)
    try:
        for _ in range(2):
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
            got = eng.get_results(track_last_interaction=False)
            assert eng.last_kernel_times()["launches"] >= 2
            _compare(got, ref, trackers=False)
    finally:
        eng.close()


def test_carry_over_with_the_consolidated_log_and_a_spawn_range(oracle):
    """enable_vpacket_tracking: the log's order is the commit order of the owners, which the carry-over must not change."""
    prob = synthetic.make_problem(seed=5, n_packets=20_000, n_shells=30, n_lines=4_000, line_interaction_type="macroatom", n_vpackets=4)
    cfg = prob.montecarlo_configuration
    cfg.ENABLE_VPACKET_TRACKING = True
    ref = _oracle(oracle, prob)
    eng = _engine(prob, variant=2, vp_carry_min_active=12)
    try:
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        _compare(got, ref)
        n = got.vpacket_log_count
        assert n == ref.vpacket_log_count and n > 0
        # (get_results hands the log over consolidated: sorted by packet and sequence number = the reference's order)
        assert np.array_equal(got.vpacket_nus[:n], ref.vpacket_nus) and np.array_equal(got.vpacket_energies[:n], ref.vpacket_energies)
        assert np.array_equal(got.vpacket_initial_mus[:n], ref.vpacket_initial_mus) and np.array_equal(got.vpacket_initial_rs[:n], ref.vpacket_initial_rs)
    finally:
        eng.close()
