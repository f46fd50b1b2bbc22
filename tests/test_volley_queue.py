"""The volley queue (variant 4): the propagation kernel requests the v-packets of a volley and suspends the packet,
vpacket_trace_kernel traces the requests of the whole grid one lane per v-packet, the next launch commits them in the
reference's order (tardis_amd/csrc/propagate_wave.hpp: VolleyRequest).  Same function of the inputs as every other
kernel: per-packet results bit-exact against the oracle, histogram and estimators to the summation-order tolerance, work
counters exact (the consolidated v-packet log: the *_nv2 / *_nv3 goldens of tests/test_hip_parity.py run on variant 4, too).

Reference behaviour held: packets/virtual_packet.py:82-386 (trace_vpacket_volley), classic/packet_propagation.py:109-118,201-244.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from tardis_amd import state as st, synthetic

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11


def _oracle(oracle, prob, **kw):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                      prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), **kw)


def _engine(prob, **options):
    from tardis_amd.engine import Engine
    eng = Engine(0)
    eng.set_option("variant", 4)
    for k, v in options.items():
        eng.set_option(k, v)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    return eng


def _compare(got, ref, trackers=True):
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    if trackers:
        for f in st.LastInteractionTrackers.I64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
        for f in st.LastInteractionTrackers.F64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL)
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.nu_bar_estimator, ref.nu_bar_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws", "packets"):
        assert got.counters[k] == ref.counters[k], k


@pytest.mark.parametrize("shape", [
    dict(n_shells=20, n_lines=3_000, line_interaction_type="downbranch", n_vpackets=10),
    dict(n_shells=60, n_lines=20_000, line_interaction_type="macroatom", n_vpackets=7),
    dict(n_shells=8, n_lines=1_000, line_interaction_type="scatter", n_vpackets=3, enable_full_relativity=True),
], ids=["downbranch-nv10", "macroatom-60shells-nv7", "scatter-fullrel-nv3"])
@pytest.mark.parametrize("min_items", [0, 4_000, -1], ids=["queue-to-the-end", "queue-then-pooled-drain", "automatic-switch"])
def test_volley_queue_matches_the_oracle(oracle, shape, min_items):
    """vq_min_items: the queue is switched off for the rest of the call once a launch requests fewer v-packets (the drain of the
    longest-lived packets then runs in one launch with the wave kernel's pooled volleys; lanes in the middle of a volley commit
    the round that came back and continue it there)."""
    prob = synthetic.make_problem(seed=5, n_packets=20_011, **shape)
    ref = _oracle(oracle, prob)
    eng = _engine(prob, vq_min_items=min_items)
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        assert eng.last_variant() == 4
        assert eng.last_kernel_times()["launches"] >= 2  # (at least: the launch volley, then the rest)
        _compare(eng.get_results(track_last_interaction=True), ref)
    eng.close()


@pytest.mark.parametrize("options", [
    {"vq_min_active": 0, "vq_min_items": 0}, {"vq_min_active": 63, "vq_min_items": 0}, {"vq_oversubscribe": 1, "waves_per_simd": 1, "vq_min_items": 0},
    {"vq_tracer_waves_per_simd": 1, "vq_min_items": 20_000}, {"vq_min_items": 200_000},
    {"log_capacity": 1 << 19, "vq_min_items": 0},  # the launches share one line-visit log: estimator passes whenever a wave's region is full
    {"log_capacity": 1 << 19, "vq_min_items": 100_000},
    {"log_capacity": 0},
    {"group_size": 16},
], ids=lambda o: "-".join(f"{k}{v}" for k, v in o.items()))
def test_volley_queue_options(oracle, options):
    """More packets than lanes (the lanes fetch new packets between volleys), every way of ending a launch, a log that fills."""
    prob = synthetic.make_problem(seed=9, n_packets=150_001, n_shells=12, n_lines=2_000, line_interaction_type="downbranch", n_vpackets=4)
    ref = _oracle(oracle, prob, track_last_interaction=False)
    eng = _engine(prob, track_last_interaction=0, **options)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    assert eng.last_variant() == 4
    _compare(eng.get_results(track_last_interaction=False), ref, trackers=False)
    eng.close()


def test_volley_queue_falls_back_where_it_does_not_apply(oracle):
    """No v-packets: nothing to queue (the wave kernel as usual); roulette with survivors: the group kernel."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=2, n_packets=5_000, n_shells=6, n_lines=500, line_interaction_type="downbranch", n_vpackets=0)
    eng = _engine(prob)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    assert eng.last_variant() in (2, 3)
    ref = _oracle(oracle, prob, track_last_interaction=False)
    got = eng.get_results(track_last_interaction=False)
    assert np.array_equal(got.output_nus, ref.output_nus)
    eng.close()
