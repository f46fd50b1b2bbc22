"""GPU parity and property tests on the table shape of BASELINE.json configs[2] -- the configuration the headline metric
is quoted on: 20 shells x 5e5 lines, macroatom line interaction (1.5e6 transitions), no v-packets.

  * a 2e4-packet problem against the CPU oracle: per-packet results bit-exact (every tracker field), estimators to 1e-11,
    work counters exact, on the kernel the engine picks by itself for this shape;
  * a 2e7-packet run (device packet source, several log-bounded chunks on two streams) through size-independent
    properties, chunking invariance, and a 1e5-packet sample against the oracle.

Reference behaviour held: macro_atom.py:52-104, modes/montecarlo_transport.py:238-373; the reference's own bar for this
loop is rtol 1e-13 (tests/test_montecarlo_main_loop.py:14-60) -- per-packet results here are bit-exact.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from tardis_amd import spectrum, state as st, synthetic

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11
SHAPE = dict(n_shells=20, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=0)
T_INNER = 1.0e4


@pytest.fixture(scope="module")
def config3():
    """Opacity tables of the configs[2] shape, resident in one engine for the whole module."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=1, n_packets=20_000, **SHAPE)
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    yield eng, prob
    eng.close()


def _oracle(oracle, prob, pc, **kw):
    return oracle.run(pc, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), **kw)


def test_config3_shape_matches_oracle(config3, oracle):
    eng, prob = config3
    pc = prob.packet_collection
    ref = _oracle(oracle, prob, pc)
    eng.set_option("variant", -1)  # the automatic choice
    eng.set_option("track_last_interaction", 1)
    eng.set_packets(pc)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    got = eng.get_results(track_last_interaction=True)
    assert np.array_equal(got.output_nus, ref.output_nus)
    assert np.array_equal(got.output_energies, ref.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
    for f in st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.nu_bar_estimator, ref.nu_bar_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws", "packets"):
        assert got.counters[k] == ref.counters[k], k
    # the macroatom walk really ran: several transitions examined per event
    assert got.counters["macro_transitions"] > 5 * got.counters["events"]
    a = spectrum.emitted_luminosity_histogram(got.output_nus, got.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    b = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    assert spectrum.relative_l2(a, b) == 0.0


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_config3_shape_every_cooperative_variant(config3, oracle, variant):
    """The group kernel (1), the wave kernel with group sweeps (2) and with lane sweeps (3) on a 4e3-packet slice."""
    eng, prob = config3
    pc = prob.packet_collection.shard(0, 5)
    ref = _oracle(oracle, prob, pc, track_last_interaction=False)
    eng.set_option("variant", variant)
    eng.set_option("track_last_interaction", 0)
    eng.set_packets(pc)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    got = eng.get_results(track_last_interaction=False)
    eng.set_option("variant", -1)
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k


def test_config3_shape_full_relativity(config3, oracle):
    """enable_full_relativity at the configs[2] table shape (set_packet_props_full_relativity, the relativistic Doppler factors,
    angle aberration, packet_propagation.py:285-318; frame_transformations.py:12-109): the engine picks the wave kernel with
    group sweeps (the lane sweeps' no-stop bounds are those of partial relativity)."""
    import copy
    eng, prob = config3
    cfg = copy.copy(prob.montecarlo_configuration)
    cfg.ENABLE_FULL_RELATIVITY = True
    pc = prob.packet_collection.shard(0, 2)
    ref = oracle.run(pc, prob.geometry, prob.time_explosion, prob.opacity_state, cfg, prob.spectrum_frequency_grid,
                     math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())
    eng.set_config(cfg, prob.spectrum_frequency_grid)
    try:
        eng.set_option("variant", -1)
        eng.set_option("track_last_interaction", 1)
        eng.set_packets(pc)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        assert eng.last_variant() == 2
    finally:
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
    for f in st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.nu_bar_estimator, ref.nu_bar_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k
    # and it is a different problem from the partial-relativity one, not a no-op switch
    part = _oracle(oracle, prob, pc)
    assert not np.array_equal(part.output_nus, ref.output_nus)


def test_config3_full_size_properties(config3, oracle):
    """2e7 packets of the configs[2] workload (a fifth of its 1e8; the bench runs the full count): every packet terminates,
    the counters add up, a different chunking reproduces the run bit for bit per packet, and the first 1e5 packets equal
    the oracle."""
    eng, prob = config3
    P = 20_000_000
    radius = float(prob.geometry.r_inner[0])
    eng.set_option("variant", -1)
    eng.set_option("track_last_interaction", 0)
    eng.create_blackbody_packets(P, radius, T_INNER)
    eng.set_option("log_capacity", 1_000_000_000)  # (1.8e9 line-visit records: at least two epochs)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    launches = eng.last_kernel_times()["launches"]
    assert launches >= 2  # log-bounded epochs (suspended and resumed lanes)
    a = eng.get_results(track_last_interaction=False)
    c = a.counters
    assert c["packets"] == P and c["events"] >= P and c["line_visits"] >= c["events"] and c["macro_transitions"] >= c["events"]
    assert c["rng_draws"] >= 2 * c["events"] - P  # a tau_event per event, a direction per interaction, a draw per jump
    assert not np.any(a.output_energies == -99.0)
    assert np.all(np.isfinite(a.output_nus)) and np.all(a.output_nus > 0)
    emitted = a.output_energies >= 0
    assert 0.005 < emitted.mean() < 0.95  # (this optically thick synthetic ejecta re-absorbs most packets)
    assert np.all(np.abs(a.output_energies) < 10.0 / P)  # Doppler factors stay within a few percent of 1 per event chain
    assert np.all(a.j_estimator > 0) and np.all(a.nu_bar_estimator > 0)
    assert np.all(a.j_blue_estimator >= 0) and np.all(a.edotlu_estimator >= 0)
    # a different split into epochs: smaller log -> more launches
    eng.set_option("log_capacity", 400_000_000)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    assert eng.last_kernel_times()["launches"] > launches
    b = eng.get_results(track_last_interaction=False)
    eng.set_option("log_capacity", 2_500_000_000)
    assert np.array_equal(a.output_nus, b.output_nus) and np.array_equal(a.output_energies, b.output_energies)
    assert a.counters == b.counters
    assert_allclose(b.j_estimator, a.j_estimator, rtol=EST_RTOL)
    assert_allclose(b.nu_bar_estimator, a.nu_bar_estimator, rtol=EST_RTOL)
    assert_allclose(b.j_blue_estimator, a.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(b.edotlu_estimator, a.edotlu_estimator, rtol=EST_RTOL)
    # the first 1e5 packets against the oracle (per-packet results do not depend on batching)
    n = 100_000
    eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=n)
    pk = eng.get_packets()
    sub = st.PacketCollection(pk["initial_radii"], pk["initial_nus"], pk["initial_mus"], pk["initial_energies"], pk["packet_seeds"],
                              4 * np.pi * st.SIGMA_SB * radius**2 * T_INNER**4)
    ref = _oracle(oracle, prob, sub, track_last_interaction=False)
    assert np.array_equal(a.output_nus[:n], ref.output_nus) and np.array_equal(a.output_energies[:n], ref.output_energies)
    ha = spectrum.emitted_luminosity_histogram(a.output_nus[:n], a.output_energies[:n], sub.time_of_simulation, prob.spectrum_frequency_grid)
    hb = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, sub.time_of_simulation, prob.spectrum_frequency_grid)
    assert spectrum.relative_l2(ha, hb) == 0.0


def test_vpackets_with_log_bounded_epochs(oracle):
    """v-packets together with a propagate call that is split into several launches by a small line-visit log: the suspended
    lanes carry the packet's volley state (v-packet sequence number, roulette predictor, look-ahead draws) across."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=31, n_packets=140_001, n_shells=6, n_lines=2_000, line_interaction_type="downbranch", n_vpackets=3)
    ref = _oracle(oracle, prob, prob.packet_collection, track_last_interaction=False)
    eng = Engine(0)
    eng.set_option("log_capacity", 1 << 19)  # 2188 waves x 256 records per epoch (the minimum)
    eng.set_option("track_last_interaction", 0)
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=False)
        assert eng.last_kernel_times()["launches"] >= 2
        assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
        assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL)
        assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
        for k in ("line_visits", "events", "vpacket_line_visits", "vpackets", "rng_draws"):
            assert got.counters[k] == ref.counters[k], k
    eng.close()
