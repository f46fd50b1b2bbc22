"""Round-5 engine options are scheduling / table-resolution choices: whatever their value, the function of the inputs is the same
(per-packet results bit-exact against the oracle, estimators and v-packet spectrum to the summation-order tolerance, counters exact).

  * `bucket_lines_permille`: resolution of the frequency-bucket index (the stopping line of a v-packet's shell crossing and a packet's
    first line start from it; csrc/tardis_mc_hip.hip) -- from 16 lines per bucket, where nearly every crossing leaves the four-line
    window and takes the four-line walks (vp_walk_to_stop, forwards and backwards), to 0.05;
  * `vpk_wide_registers`: the two-waves-per-SIMD instantiation of the v-packet kernels, never / automatic / always;
  * `vpk_wave_min_packets`: wave-owner kernel or group kernel for v-packet calls on fine grids;
  * `pass_cus`: CU-masked streams for the propagation launches and the estimator passes of a multi-epoch call (tests/test_full_size_configs.py
    runs it at 3e7 packets).

Reference behaviour held: packets/virtual_packet.py:82-386, modes/montecarlo_transport.py:238-373.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from tardis_amd import synthetic

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11


def _oracle(oracle, prob):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), track_last_interaction=False)


def _run(prob, **options):
    from tardis_amd.engine import Engine
    with Engine(0) as eng:
        eng.set_option("track_last_interaction", 0)
        for k, v in options.items():
            eng.set_option(k, v)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        return eng.get_results(track_last_interaction=False), eng.last_variant()


def _same(got, ref):
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL, atol=0)
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k


@pytest.fixture(scope="module")
def thick(oracle):
    """100 shells x 3e5 lines (screened: n_lines >= 2500 n_shells), ten v-packets, heavy-tailed blocks; 4 000 packets."""
    prob = synthetic.make_problem(seed=17, n_packets=4_000, n_shells=100, n_lines=300_000, line_interaction_type="macroatom", n_vpackets=10,
                                  level_sizes="heavy")
    return prob, _oracle(oracle, prob)


@pytest.fixture(scope="module")
def thin(oracle):
    """The tardis_example shape with v-packets (no screening: every v-packet is traced line by line, most leave the grid alive)."""
    prob = synthetic.make_problem(seed=18, n_packets=30_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=5)
    return prob, _oracle(oracle, prob)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("permille", [16_000, 3_000, 750, 50])
def test_bucket_index_resolution_thick(thick, variant, permille):
    prob, ref = thick
    got, v = _run(prob, variant=variant, bucket_lines_permille=permille)
    assert v == variant
    _same(got, ref)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("permille", [16_000, 50])
def test_bucket_index_resolution_thin(thin, variant, permille):
    prob, ref = thin
    got, v = _run(prob, variant=variant, bucket_lines_permille=permille)
    assert v == variant
    _same(got, ref)


@pytest.mark.parametrize("wide", [0, 1, 2])
def test_wide_register_instantiations(thick, thin, wide):
    for prob, ref in (thick, thin):
        got, v = _run(prob, variant=2, vpk_wide_registers=wide)
        assert v == 2
        _same(got, ref)


def test_kernel_choice_threshold(thick):
    prob, ref = thick
    got, v = _run(prob, vpk_wave_min_packets=1_000)   # 4 000 packets: the wave-owner kernel
    assert v == 2
    _same(got, ref)
    got, v = _run(prob)                               # the default threshold (1e5): the group kernel
    assert v == 1
    _same(got, ref)


# ---- the two instantiations of the lane-sweep kernel (option ls_waves_per_simd: 4 = 128 VGPRs, sixteen waves per CU, eight lines per step; 3 = 166
# VGPRs, twelve waves per CU, twelve lines per step; 0 = the engine times both on the first calls of a key and keeps the faster)

LS_PROBLEMS = [
    dict(seed=5, n_packets=40_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch"),
    dict(seed=6, n_packets=20_000, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", level_sizes="heavy"),
    dict(seed=7, n_packets=30_000, n_shells=5, n_lines=37, line_interaction_type="scatter"),      # a line list shorter than three chunks
    dict(seed=8, n_packets=30_000, n_shells=8, n_lines=12, line_interaction_type="downbranch"),   # ... than one twelve-line chunk
]


def _oracle_full(oracle, prob):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())


@pytest.mark.parametrize("wps", [3, 4])
@pytest.mark.parametrize("kw", LS_PROBLEMS, ids=["tardis_example", "config3-heavy", "37-lines", "12-lines"])
def test_lane_sweep_instantiations_match_the_oracle(oracle, kw, wps):
    from tardis_amd import state as st
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(**kw)
    ref = _oracle_full(oracle, prob)
    with Engine(0) as eng:
        eng.set_option("ls_waves_per_simd", wps)
        eng.set_option("log_capacity", 1 << 19)  # (several launches: lanes are suspended in the middle of twelve-line sweeps and resumed)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        assert eng.last_variant() == 3
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


def test_the_engine_choosing_between_them_never_changes_a_result(oracle):
    """Default (ls_waves_per_simd 0): the first call of a (packet count, tables) key runs instantiation A untimed, then A, B, A, B timed, from
    the sixth on B if its faster call beat A's faster one by 3 % and A otherwise -- eight calls, eight identical results."""
    from tardis_amd.engine import Engine
    # (more packets than four fills of the grid's lanes: below that the engine takes B without timing anything)
    prob = synthetic.make_problem(seed=9, n_packets=1_200_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    ref = _oracle_full(oracle, prob)
    with Engine(0) as eng:
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        for call in range(8):
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
            got = eng.get_results(track_last_interaction=False)
            assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies), call
            assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
            assert got.counters["line_visits"] == ref.counters["line_visits"] and got.counters["rng_draws"] == ref.counters["rng_draws"]
