"""GPU parity of the hot sectors of the macro-atom walk (tardis_amd/csrc/walk_tables.hpp, round 4).

macro_atom_interaction (tardis/transport/montecarlo/macro_atom.py:52-104) selects the first transition of the activated
level's block whose running probability sum exceeds the number drawn.  A block with a hot sector is entered through ONE
64-byte record holding its six widest probability intervals in 16-bit units; a number that falls into one of them is
decided there (`lo <= x < hi` implies the reference's choice), any other number is looked up again -- the same number, no new
draw -- in the block's own tables.  Held here:

  * every golden case with a macro atom, with no block / every block / the automatic choice of blocks on hot sectors:
    per-packet results, trackers, estimators and work counters as in test_hip_parity.py (`macro_transitions` is the
    reference's count of examined rows: a hot entry carries the row it stands for);
  * heavy-tailed blocks at the configs[2] table shape (the workload hot sectors exist for): bit-exact against the oracle,
    and device counters show that both outcomes of a probe -- decided / looked up again -- happened, thousands of times;
  * a number that is looked up again survives a carried-over walk and the suspension of its wave (tiny line-visit log:
    many epochs) -- same results as one launch.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st, synthetic
from test_hip_parity import _check_golden_case

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11
MACRO_CASES = [n for n in _golden.CASES if "macroatom" in n or "downbranch" in n]
HOT_HIT, HOT_MISS = 65536, 131072  # debug flags -> counters["reserved"]


@pytest.fixture(scope="module")
def engine():
    from tardis_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize("walk_hot", [0, 1, -1])
@pytest.mark.parametrize("name", MACRO_CASES)
def test_golden_cases_with_and_without_hot_sectors(engine, oracle, name, walk_hot):
    engine.set_option("walk_hot", walk_hot)  # (takes effect in the set_opacity of the call)
    if walk_hot == -1:
        engine.set_option("walk_hot_min_mass", 300)  # small golden blocks: some hot, some not
    try:
        for variant in ((-1, 2, 3) if "_nv" not in name or "_nv0" in name else (-1,)):
            engine.set_option("variant", variant)
            _check_golden_case(engine, oracle, name)
    finally:
        engine.set_option("variant", -1)
        engine.set_option("walk_hot", -1)
        engine.set_option("walk_hot_min_mass", 800)


def _oracle(oracle, prob, pc, **kw):
    return oracle.run(pc, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), **kw)


def _run(eng, pc, track, flags=0):
    eng.set_option("debug_flags", flags)
    eng.set_option("track_last_interaction", int(track))
    eng.set_packets(pc)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    eng.set_option("debug_flags", 0)
    return eng.get_results(track_last_interaction=track)


@pytest.fixture(scope="module", params=["macroatom", "downbranch"])
def heavy(request):
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=11, n_packets=16_000, n_shells=20, n_lines=500_000, line_interaction_type=request.param,
                                  level_sizes="heavy")
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    yield eng, prob, request.param
    eng.close()


@pytest.mark.parametrize("walk_hot", [1, -1])
def test_heavy_blocks_through_hot_sectors(heavy, oracle, walk_hot):
    eng, prob, mode = heavy
    eng.set_option("walk_hot", walk_hot)
    eng.set_opacity(prob.opacity_state)
    pc = prob.packet_collection
    ref = _oracle(oracle, prob, pc)
    got = _run(eng, pc, True)
    assert eng.last_variant() == 3
    assert np.array_equal(got.output_nus, ref.output_nus)
    assert np.array_equal(got.output_energies, ref.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
    for f in st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws", "packets"):
        assert got.counters[k] == ref.counters[k], k
    # both outcomes of a probe were exercised (and the results above did not change with the counters on)
    sub = pc.shard(0, 4)
    ref_s = _oracle(oracle, prob, sub, track_last_interaction=False)
    hits = _run(eng, sub, False, flags=HOT_HIT)
    assert np.array_equal(hits.output_nus, ref_s.output_nus) and np.array_equal(hits.output_energies, ref_s.output_energies)
    misses = _run(eng, sub, False, flags=HOT_MISS)
    assert np.array_equal(misses.output_nus, ref_s.output_nus)
    n_hit, n_miss = hits.counters["reserved"], misses.counters["reserved"]
    jumps_upper = ref_s.counters["rng_draws"] - ref_s.counters["events"]
    assert n_hit > 1000 and n_miss > 100 and n_hit + n_miss <= jumps_upper, (n_hit, n_miss, jumps_upper)
    if walk_hot == -1:
        assert n_hit > 3 * n_miss, (n_hit, n_miss)  # the automatic choice keeps blocks whose sector decides little off it
    eng.set_option("walk_hot", -1)


def test_hot_sectors_off_is_the_round3_walk(heavy, oracle):
    eng, prob, mode = heavy
    eng.set_option("walk_hot", 0)
    eng.set_opacity(prob.opacity_state)
    sub = prob.packet_collection.shard(1, 4)
    ref = _oracle(oracle, prob, sub, track_last_interaction=False)
    got = _run(eng, sub, False, flags=HOT_HIT | HOT_MISS)
    assert got.counters["reserved"] == 0
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert got.counters["macro_transitions"] == ref.counters["macro_transitions"]
    eng.set_option("walk_hot", -1)


@pytest.mark.parametrize("variant", [2, 3])
def test_looked_up_again_numbers_survive_carried_walks_and_epochs(oracle, variant):
    """Every block on a hot sector (most probes of these 12..24-row blocks miss: the number is looked up again), walks carried
    over from pass to pass, and a line-visit log so small that the waves suspend again and again: one launch and many epochs
    agree bit for bit, and with the oracle."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=23, n_packets=(1 << 19) + 333, n_shells=6, n_lines=3_000, line_interaction_type="macroatom")
    ref = _oracle(oracle, prob, prob.packet_collection)
    outs = []
    for cap in (1_500_000_000, 1_200_000):
        eng = Engine(0)
        eng.set_option("variant", variant)
        eng.set_option("walk_hot", 1)
        eng.set_option("log_capacity", cap)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        outs.append(eng.get_results(track_last_interaction=True))
        launches = eng.last_kernel_times()["launches"]
        assert launches == 1 if cap > 1_000_000_000 else launches >= 3
        eng.close()
    a, b = outs
    assert np.array_equal(a.output_nus, ref.output_nus) and np.array_equal(a.output_energies, ref.output_energies)
    assert np.array_equal(a.output_nus, b.output_nus) and np.array_equal(a.output_energies, b.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS + st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(a.trackers, f), getattr(b.trackers, f), equal_nan=True), f
    assert_allclose(a.j_blue_estimator, b.j_blue_estimator, rtol=EST_RTOL)
    assert a.counters == b.counters
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert a.counters[k] == ref.counters[k], k
