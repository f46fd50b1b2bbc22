"""v-packet screening (tardis_amd/csrc/tau_prefix.hpp): with the default survival probability 0 a v-packet whose optical depth passes
tau_russian is dropped whatever the depth was (trace_vpacket, packets/virtual_packet.py:179-244), so the engine decides that from
prefix sums of tau -- with a rigorous error margin -- and traces line by line only the v-packets that leave the grid alive or come
within the margin of the threshold.  Everything observable must be what the line-by-line trace gives: per-packet outputs (the
roulette draws shift the parent's stream), the v-packet histogram and log, the work counters (`vpacket_line_visits` counts the
lines the reference's loop would have visited)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import synthetic

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11
DECIDED = 67108864  # debug flag: counters["reserved"] >> 40 = v-packets decided on the prefix sums


def _oracle(oracle, prob):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), track_last_interaction=False)


def _run(eng, prob, variant, screening, flags=0):
    eng.set_option("variant", variant)
    eng.set_option("vpacket_screening", screening)
    eng.set_option("debug_flags", flags)
    eng.set_option("track_last_interaction", 0)
    try:
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        return eng.get_results(track_last_interaction=False)
    finally:
        eng.set_option("variant", -1); eng.set_option("vpacket_screening", -1); eng.set_option("debug_flags", 0)


@pytest.fixture(scope="module")
def engine():
    from tardis_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


def _same(got, ref, log):
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL, atol=0)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k
    if log:
        n = got.vpacket_log_count
        assert n == ref.vpacket_log_count
        assert np.array_equal(got.vpacket_nus[:n], ref.vpacket_nus) and np.array_equal(got.vpacket_energies[:n], ref.vpacket_energies)
        assert np.array_equal(got.vpacket_initial_mus[:n], ref.vpacket_initial_mus)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name", [n for n in _golden.CASES if ("_nv2" in n or "_nv3" in n or "_nv10" in n) and "roulette" not in n])
def test_screening_forced_on_reproduces_the_vpacket_goldens(engine, oracle, name, variant):
    prob, g = _golden.load_case(name)
    ref = _oracle(oracle, prob)
    got = _run(engine, prob, variant, 1)
    _same(got, ref, "vpacket_nus" in g)
    assert_allclose(got.v_packets_energy_hist, g["v_packets_energy_hist"], rtol=EST_RTOL, atol=0)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("mode,full", [("macroatom", False), ("downbranch", True)])
def test_screening_on_thick_ejecta_decides_most_vpackets(engine, oracle, mode, full, variant):
    """Optically thick lines on a fine grid: most v-packets are dropped by the roulette, a few leave the grid alive (and are traced
    line by line after the screening did not decide them); the v-packet log records every one of them."""
    prob = synthetic.make_problem(seed=41, n_packets=3000, n_shells=40, n_lines=20_000, line_interaction_type=mode, n_vpackets=4,
                                  log_tau_mean=-2.0, enable_full_relativity=full)
    prob.montecarlo_configuration.ENABLE_VPACKET_TRACKING = True
    ref = _oracle(oracle, prob)
    dropped = int((ref.vpacket_energies == 0).sum())
    alive = int((ref.vpacket_energies > 0).sum())
    assert dropped > 10 * max(alive, 1) and alive > 20  # (the problem has both kinds)
    got = _run(engine, prob, variant, 1, flags=DECIDED)
    decided = got.counters["reserved"] >> 40
    _same(got, ref, True)
    assert 0.5 * dropped < decided <= 2 * dropped  # (discarded speculative traces count, too; a packet's first volley is not screened)
    off = _run(engine, prob, variant, 0, flags=DECIDED)
    assert off.counters["reserved"] >> 40 == 0
    _same(off, ref, True)


def test_screening_is_off_with_a_survival_probability_or_negative_optical_depths(engine, oracle):
    prob = synthetic.make_problem(seed=42, n_packets=800, n_shells=30, n_lines=8_000, line_interaction_type="downbranch", n_vpackets=3,
                                  log_tau_mean=-1.5)
    prob.opacity_state.tau_sobolev[17, :] = -1e-3  # (a negative optical depth: the prefix sums would not bound the serial sum)
    ref = _oracle(oracle, prob)
    got = _run(engine, prob, 1, 1, flags=DECIDED)
    assert got.counters["reserved"] >> 40 == 0
    _same(got, ref, False)
    prob = synthetic.make_problem(seed=43, n_packets=800, n_shells=30, n_lines=8_000, line_interaction_type="downbranch", n_vpackets=3,
                                  log_tau_mean=-1.5)
    prob.montecarlo_configuration.SURVIVAL_PROBABILITY = 0.25
    ref = _oracle(oracle, prob)
    got = _run(engine, prob, -1, 1, flags=DECIDED)
    assert got.counters["reserved"] >> 40 == 0
    _same(got, ref, False)


@pytest.mark.parametrize("variant", [1, 2])
def test_optical_depth_exactly_at_the_threshold_is_never_decided_on_prefix_sums(engine, oracle, variant):
    """The margin of the screening, attacked (VERDICT r03 weak-1 i).  Every line of every shell has tau = 0.25 and electron
    scattering is switched off (sigma_T = 1e-200: chi d is absorbed by the sum), so a v-packet's running depth is an exact
    multiple of 0.25 in the reference's serial sum AND in the prefix sums -- and equals VPACKET_TAU_RUSSIAN = 10 exactly whenever
    40 lines lie behind it at a shell boundary.  The reference's test there is `tau > tau_russian` (virtual_packet.py:219-232):
    false at 10.0, true one ulp below.  Three thresholds -- 10, the double below, the double above -- give three reference runs
    of which the first two DIFFER (the ties flip: they consume a roulette draw and die) while the last two agree.  The screening
    cannot tell them apart (one ulp is far inside its margin), so it must leave every tie to the line-by-line trace: the engine
    reproduces each run, and the number of v-packets it decided on prefix sums is the same for all three."""
    prob = synthetic.make_problem(seed=47, n_packets=1500, n_shells=30, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=3)
    prob.opacity_state.tau_sobolev[:, :] = 0.25
    cfg = prob.montecarlo_configuration
    cfg.DISABLE_ELECTRON_SCATTERING = True
    cfg.ENABLE_VPACKET_TRACKING = True
    thresholds = [10.0, float(np.nextafter(10.0, 0.0)), float(np.nextafter(10.0, 20.0))]
    refs, decided = [], []
    for thr in thresholds:
        cfg.VPACKET_TAU_RUSSIAN = thr
        ref = _oracle(oracle, prob)
        got = _run(engine, prob, variant, 1, flags=DECIDED)
        # (per-packet outputs, histogram, counters: a single mis-decided tie shifts its parent's stream and shows in all of them;
        # the consolidated log -- 4e5 entries here -- is compared on the smaller problems above)
        _same(got, ref, False)
        refs.append(ref)
        decided.append(got.counters["reserved"] >> 40)
        off = _run(engine, prob, variant, 0)
        _same(off, ref, False)
    at, below, above = refs
    n = at.vpacket_log_count
    assert n == above.vpacket_log_count and np.array_equal(at.vpacket_energies, above.vpacket_energies) and at.counters == above.counters
    # the ties exist, many of them: v-packets alive at tau == 10.0 that one ulp kills (their draw then shifts the parent's stream, so
    # the runs diverge altogether: compare what cannot be shifted, the totals)
    assert below.counters["rng_draws"] != at.counters["rng_draws"]
    flipped = int((at.vpacket_energies[: min(n, below.vpacket_log_count)] != below.vpacket_energies[: min(n, below.vpacket_log_count)]).sum())
    assert flipped > 50, flipped
    assert decided[0] > 1000, decided  # the screening was at work ...
    assert decided[0] == decided[2], decided  # ... and decided the same v-packets whichever side of the tie the threshold is on
    # (below the tie the runs diverge after the first flipped v-packet, so its count is a different run's: only sanity)
    assert decided[1] > 1000, decided
