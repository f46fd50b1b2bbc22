"""Round-6 engine options change where the lane sweep reads its lines from, never what a trace decides.

  * `sweep_table` 1 / 2: the sweep of `trace_packet` (modes/homologous_rad_packet_transport.py:100-155) reads the interleaved table
    nt_t[shell][line] = {nu_line, tau_sobolev} (csrc/tardis_mc_hip.hip: interleave_kernel; propagate_wave_kernel<..., NT>) instead of the
    line list and the shell's tau row -- runs from the current line (1) or whole aligned 128-byte runs with the entries in front of the
    current line skipped (2).  Same operands per line, same operations: per-packet results bit-exact against the oracle and against the
    reference-generated goldens, estimators to the summation-order tolerance, counters exact -- on line counts that are and are not
    multiples of the eight-line runs, on lists shorter than one run, with lanes suspended in the middle of a sweep (several launches).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st
from tardis_amd import synthetic

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11

PROBLEMS = [
    dict(seed=5, n_packets=40_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch"),
    dict(seed=6, n_packets=20_000, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", level_sizes="heavy"),
    dict(seed=7, n_packets=30_000, n_shells=5, n_lines=37, line_interaction_type="scatter"),       # L % 8 = 5: rows padded to 40 entries
    dict(seed=8, n_packets=30_000, n_shells=8, n_lines=12, line_interaction_type="downbranch"),    # shorter than two runs
    dict(seed=9, n_packets=30_000, n_shells=3, n_lines=5, line_interaction_type="macroatom"),      # shorter than one run
    dict(seed=10, n_packets=20_000, n_shells=20, n_lines=30_003, line_interaction_type="macroatom"),
]
IDS = ["tardis_example", "config3-heavy", "37-lines", "12-lines", "5-lines", "30003-lines"]


def _oracle_full(oracle, prob):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())


def _check(got, ref):
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


@pytest.mark.parametrize("table,wps", [(1, 4), (2, 4), (1, 3)])
@pytest.mark.parametrize("kw", PROBLEMS, ids=IDS)
def test_interleaved_sweep_table_matches_the_oracle(oracle, kw, table, wps):
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(**kw)
    ref = _oracle_full(oracle, prob)
    with Engine(0) as eng:
        eng.set_option("sweep_table", table)
        eng.set_option("ls_waves_per_simd", wps)
        eng.set_option("log_capacity", 1 << 19)  # (several launches: lanes are suspended in the middle of a sweep and resumed)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        assert eng.last_variant() == 3
        _check(got, ref)
        # a second opacity state on the same context: the table is rebuilt (here: the same lines with every optical depth doubled)
        op2 = prob.opacity_state
        op2 = st.OpacityState(op2.electron_density, op2.t_electrons, op2.line_list_nu, 2.0 * op2.tau_sobolev, op2.transition_probabilities,
                              op2.line2macro_level_upper, op2.macro_block_edge_index, op2.transition_type, op2.destination_level_id,
                              op2.transition_line_id)
        ref2 = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, op2, prob.montecarlo_configuration,
                          prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())
        eng.set_opacity(op2)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        _check(eng.get_results(track_last_interaction=True), ref2)


@pytest.mark.parametrize("table", [1, 2])
@pytest.mark.parametrize("name", [n for n in _golden.CASES if "_nv" not in n or "_nv0" in n])
def test_interleaved_sweep_table_on_the_goldens(oracle, name, table):
    """The reference-generated fixtures without v-packets (the lane sweeps' domain; full-relativity cases take the group sweeps and
    ignore the option)."""
    from tardis_amd.engine import Engine
    prob, g = _golden.load_case(name)
    with Engine(0) as eng:
        eng.set_option("sweep_table", table)
        eng.set_option("ls_waves_per_simd", 4)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
    assert_allclose(got.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(got.output_energies, g["output_energies"], rtol=1e-13, atol=0)
    for f in _golden.TRACKER_I64:
        assert np.array_equal(getattr(got.trackers, f), g["trk_" + f]), f
    stride = int(g["line_estimator_stride"])
    assert_allclose(got.j_blue_estimator[::stride], g["j_blue_estimator"], rtol=EST_RTOL, atol=0)


def test_a_negative_optical_depth_keeps_the_general_proof(oracle):
    """The lean no-stop proof of the interleaved-table kernels assumes tau >= 0 (an electron-scattering stop is excluded through the line's own
    optical depth).  A table with negative entries -- population inversions can produce them -- is detected when the interleaved table is built and
    such a problem stays on the separate tables with the five-test proof: same results as the oracle, whatever `sweep_table` asks for."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=11, n_packets=30_000, n_shells=12, n_lines=20_000, line_interaction_type="macroatom")
    op = prob.opacity_state
    tau = op.tau_sobolev.copy()
    rng = np.random.default_rng(5)
    idx = rng.integers(0, tau.size, 400)
    tau.reshape(-1)[idx] = -np.abs(tau.reshape(-1)[idx]) * 0.5 - 1e-3   # 400 negative optical depths, some of them large
    bad = st.OpacityState(op.electron_density, op.t_electrons, op.line_list_nu, tau, op.transition_probabilities, op.line2macro_level_upper,
                          op.macro_block_edge_index, op.transition_type, op.destination_level_id, op.transition_line_id)
    ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, bad, prob.montecarlo_configuration, prob.spectrum_frequency_grid,
                     math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads())
    for table in (-1, 1, 2):
        with Engine(0) as eng:
            eng.set_option("sweep_table", table)
            eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(bad)
            eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
            got = eng.get_results(track_last_interaction=True)
            assert eng.last_variant() == 3
        assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies), table
        for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
            assert got.counters[k] == ref.counters[k], (table, k)
        assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
