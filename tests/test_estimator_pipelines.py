"""GPU parity of the two ways an epoch's line-visit log becomes j_blue / Edotlu (round 4).

update_line_estimators (tardis/transport/montecarlo/estimators/estimators_line.py) adds, for every line a packet passes, a term
to j_blue[line, shell] and Edotlu[line, shell].  The engine logs one record per trace and adds the terms afterwards:

  * est_pipeline 1 (default; csrc/estimator_partition.hpp): the records are grouped by shell and then by (shell, 2048-line tile)
    in two LDS-staged partition passes, and accumulate_blocks_kernel reads them in order -- one add per line before the first and
    after the last 8-line boundary of a trace, one add per aligned 8-line block in between;
  * est_pipeline 0 (csrc/estimator_log.hpp): an index of the records is counting-sorted by bin and the records are fetched through
    it; est_accumulate 1 / 0 = the same kernel with block sums / round 3's one-add-per-visit kernel.

All of them must give the reference's sums (a different association of the same terms: 1e-11, the tolerance the estimators are held
to everywhere), leave lines nobody visited at exactly 0, and leave everything else -- per-packet results, counters -- untouched.
Covered here beyond the goldens (the whole GPU suite runs on the default pipeline): traces longer than the 255 lines the pooled path
takes and traces that run past their tile's apron (checked on the oracle's trace log: the problem has them); many epochs with odd
chunk sizes; a log whose shells hold so few records that a staged segment spans more buckets than the partition kernel keeps in LDS
(its record-by-record fallback); full relativity (the flush applies no nu_line factor).
"""
import ctypes as C

import numpy as np
import pytest
from numpy.testing import assert_allclose

from tardis_amd import synthetic

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11
MODES = [(1, 1), (1, 2), (0, 1), (0, 2), (0, 0), (1, 3), (0, 3)]  # (est_pipeline, est_accumulate); accumulate 2 / 3 = the dyadic hierarchy of round 6: items spread over the lanes / a lane per record


def _oracle(oracle, prob, trace_log=None):
    lib = oracle.lib()
    if trace_log is not None:
        lib.oracle_set_trace_log.restype = None
        lib.oracle_set_trace_log.argtypes = [C.c_void_p, C.c_int64]
        lib.oracle_trace_log_count.restype = C.c_int64
        lib.oracle_set_trace_log(trace_log.ctypes.data, len(trace_log))
    try:
        ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                         prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=1 if trace_log is not None else oracle.max_threads(),
                         track_last_interaction=False)
        n = int(lib.oracle_trace_log_count()) if trace_log is not None else 0
    finally:
        if trace_log is not None:
            lib.oracle_set_trace_log(None, 0)
    return ref, n


def _run(prob, pipeline, accumulate, **options):
    from tardis_amd.engine import Engine
    eng = Engine(0)
    try:
        eng.set_option("est_pipeline", pipeline)
        eng.set_option("est_accumulate", accumulate)
        eng.set_option("track_last_interaction", 0)
        for k, v in options.items():
            eng.set_option(k, v)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        return eng.get_results(track_last_interaction=False), eng.last_kernel_times()["launches"], eng.last_variant()
    finally:
        eng.close()


def _same(got, ref):
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL, atol=0)
    assert np.array_equal(got.j_blue_estimator == 0.0, ref.j_blue_estimator == 0.0)  # unvisited lines: exactly zero
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k


@pytest.fixture(scope="module")
def long_traces(oracle):
    """Dense line list, thin lines: traces of hundreds of lines, some past the 256-line apron of their tile."""
    prob = synthetic.make_problem(seed=5, n_packets=12_000, n_shells=8, n_lines=400_000, line_interaction_type="macroatom", log_tau_mean=-5.0)
    log = np.zeros((4_000_000, 4), dtype=np.int64)
    ref, n = _oracle(oracle, prob, log)
    tr = log[:n]
    start, cnt = tr[:, 1], tr[:, 2]
    assert (cnt > 255).sum() > 100, "no trace for the whole-wave path"
    past_apron = (start % 2048 + cnt > 2048 + 256).sum()
    assert past_apron > 20, "no trace past its tile's apron"
    assert ((cnt > 0) & (cnt < 8)).sum() > 100 and (cnt % 8 == 0).sum() > 100  # heads / tails only, whole blocks only
    return prob, ref


@pytest.mark.parametrize("pipeline,accumulate", MODES)
def test_long_traces_and_aprons(long_traces, pipeline, accumulate):
    prob, ref = long_traces
    got, launches, variant = _run(prob, pipeline, accumulate)
    assert variant in (2, 3) and launches == 1
    _same(got, ref)


@pytest.mark.parametrize("pipeline,accumulate", MODES)
def test_many_epochs_and_odd_chunks(long_traces, pipeline, accumulate):
    """A log of half of what the call writes, in chunks of 257 records: several epochs, segments of odd length."""
    prob, ref = long_traces
    got, launches, _ = _run(prob, pipeline, accumulate, log_capacity=200_000, log_chunk_records=257)
    assert launches >= 2
    _same(got, ref)


@pytest.mark.parametrize("sets", [1, 2])
def test_passes_before_or_beside_the_next_launch(long_traces, sets):
    """`log_sets` 1: one log set, the passes of an epoch run before the next launch (what a call of several epochs takes by itself); 2: two sets, the
    passes run on a second stream beside the next launch.  Both on a log a fifth of what the call writes."""
    prob, ref = long_traces
    got, launches, _ = _run(prob, 1, 3, log_sets=sets, log_capacity=100_000, log_chunk_records=512)
    assert launches >= 4
    _same(got, ref)


@pytest.mark.parametrize("one_level", [1, 0])
@pytest.mark.parametrize("options", [dict(), dict(log_capacity=300_000, log_chunk_records=300)], ids=["one-launch", "epochs"])
def test_few_bins_take_one_partition_pass(oracle, one_level, options):
    """Short line lists (3e4 lines: 15 tiles x 20 shells = 300 bins, the tardis_example shape) rank all bins of a staged segment in LDS at once: one partition
    pass, log chunks -> scratch copy by bin, instead of two (`est_one_level` 0 keeps the two).  Same sums."""
    prob = synthetic.make_problem(seed=17, n_packets=60_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    ref, _ = _oracle(oracle, prob)
    for accumulate in (3, 1):
        got, launches, variant = _run(prob, 1, accumulate, est_one_level=one_level, **options)
        assert variant == 3 and (launches >= 3 if options else launches == 1)
        _same(got, ref)


def test_two_sets_inside_the_buffers_of_a_larger_one_set_call(oracle, long_traces):
    """A context that has run a larger call on ONE log set places the two sets of a later, smaller call inside that set's buffers (no second allocation of
    tens of GB in the middle of a run): the smaller call -- many epochs, the passes of one beside the launch of the next -- still gives the oracle's sums."""
    from tardis_amd.engine import Engine
    prob, ref = long_traces
    big = synthetic.make_problem(seed=6, n_packets=120_000, n_shells=8, n_lines=400_000, line_interaction_type="macroatom", log_tau_mean=-5.0)
    with Engine(0) as eng:
        eng.set_option("track_last_interaction", 0)
        eng.set_option("log_sets", 1)
        eng.set_geometry(big.geometry, big.time_explosion); eng.set_opacity(big.opacity_state)
        eng.set_config(big.montecarlo_configuration, big.spectrum_frequency_grid); eng.set_packets(big.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        eng.set_option("log_sets", 2); eng.set_option("log_capacity", 150_000); eng.set_option("log_chunk_records", 512)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=False)
        assert eng.last_kernel_times()["launches"] >= 3
    _same(got, ref)


@pytest.mark.parametrize("accumulate", [1, 2])
@pytest.mark.parametrize("options", [dict(), dict(log_capacity=200_000, log_chunk_records=257)], ids=["one-launch", "epochs-odd-chunks"])
def test_two_level_partition_without_the_shell_sorted_log(long_traces, accumulate, options):
    """Round 6: the production lane-sweep kernels write a shell-sorted log (one open chunk per shell and wave) and the passes start with the
    partition by bin -- what the tests above run.  `log_by_shell` 0 keeps round 4's form: chunks of mixed shells, partition by shell, then by bin."""
    prob, ref = long_traces
    got, launches, variant = _run(prob, 1, accumulate, log_by_shell=0, **options)
    assert variant == 3
    _same(got, ref)


def test_shell_sorted_log_with_more_shells_than_a_pass_has_lanes_in_one(oracle):
    """64 shells (the most the shell-sorted log takes), few packets: nearly every record of a pass goes to a shell of its own, chunks close
    nearly empty, and a pool of a few chunks per wave runs dry again and again (many epochs)."""
    prob = synthetic.make_problem(seed=21, n_packets=6_000, n_shells=64, n_lines=40_000, line_interaction_type="downbranch")
    ref, _ = _oracle(oracle, prob)
    for options in (dict(), dict(log_capacity=60_000, log_chunk_records=256)):
        got, launches, variant = _run(prob, 1, 2, **options)
        assert variant == 3
        _same(got, ref)


@pytest.mark.parametrize("full", [False, True])
def test_sparse_shells_take_the_record_by_record_fallback(oracle, full):
    """100 shells x 3e5 lines (147 bins per shell) and so few packets that a shell logs ~100 records: the 2048 records a workgroup of the
    second partition pass stages span ~20 shells = 3000 bins > the 1024 it ranks in LDS."""
    prob = synthetic.make_problem(seed=9, n_packets=150, n_shells=100, n_lines=300_000, line_interaction_type="downbranch",
                                  enable_full_relativity=full)
    ref, _ = _oracle(oracle, prob)
    for pipeline, accumulate in MODES[:2]:
        got, _, _ = _run(prob, pipeline, accumulate, variant=3)
        _same(got, ref)


def test_pipelines_agree_to_rounding_on_heavy_tables(oracle):
    """Same call, the three ways: the sums differ by association only (<< the tolerance against the oracle)."""
    prob = synthetic.make_problem(seed=13, n_packets=30_000, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", level_sizes="heavy")
    outs = [_run(prob, p, a)[0] for p, a in MODES]
    ref, _ = _oracle(oracle, prob)
    for got in outs:
        _same(got, ref)
    for got in outs[1:]:
        assert_allclose(got.j_blue_estimator, outs[0].j_blue_estimator, rtol=1e-13, atol=0)
        assert_allclose(got.edotlu_estimator, outs[0].edotlu_estimator, rtol=1e-13, atol=0)
