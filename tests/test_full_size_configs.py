"""BASELINE.json's configurations at THEIR OWN packet counts (GPU): configs[2] -- 20 shells x 5e5 lines, macroatom, 1e8 packets,
the workload `bench.py` quotes the headline on -- and configs[4]'s per-GPU share -- 100 shells x 5e5 lines, macroatom, ten
v-packets per interaction, 5e8 / 8 = 6.25e7 packets.  The smaller tests (tests/test_config3_shape.py, test_hip_parity.py) hold
the same table shapes against the oracle on every packet; at the full counts the oracle checks a sample (per-packet results do
not depend on batching) and the rest is size-independent properties: every packet terminates, the counters add up, a
different split of the call into epochs reproduces every packet bit for bit.

Reference behaviour held: modes/montecarlo_transport.py:238-373 (main loop), macro_atom.py:52-104, packets/virtual_packet.py:82-386.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from tardis_amd import spectrum, state as st, synthetic

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11
T_INNER = 1.0e4


def _sample(eng, oracle, prob, P, n, radius):
    """The first n packets of the P-packet device draw, on the CPU oracle."""
    eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=n)
    pk = eng.get_packets()
    sub = st.PacketCollection(pk["initial_radii"], pk["initial_nus"], pk["initial_mus"], pk["initial_energies"], pk["packet_seeds"],
                              4 * np.pi * st.SIGMA_SB * radius**2 * T_INNER**4)
    ref = oracle.run(sub, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                     prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(),
                     track_last_interaction=False)
    return sub, ref


def test_baseline_config3_at_its_own_1e8_packets(oracle):
    """BASELINE configs[2] as bench.py runs it: heavy-tailed macro-atom blocks, tracking on, device packet source, 1e8 packets in
    one propagate call (several epochs over one packet supply)."""
    from tardis_amd.engine import Engine
    P = 100_000_000
    kw = dict(synthetic.BASELINE_CONFIGS[3])
    assert int(kw.pop("n_packets")) == P and kw["n_lines"] == 500_000 and kw["n_shells"] == 20
    prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes="heavy", **kw)
    radius = float(prob.geometry.r_inner[0])
    eng = Engine(0)
    try:
        eng.set_option("track_last_interaction", 1)
        eng.set_geometry(prob.geometry, prob.time_explosion)
        eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        eng.create_blackbody_packets(P, radius, T_INNER)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        assert eng.last_variant() == 3  # wave kernel, lane sweeps
        launches = eng.last_kernel_times()["launches"]
        assert launches >= 3  # (the line-visit log of 1e8 packets does not fit one epoch)
        a = eng.get_results(track_last_interaction=False)
        c = a.counters
        assert c["packets"] == P and c["events"] >= P and c["line_visits"] >= c["events"] and c["macro_transitions"] >= c["events"]
        assert c["rng_draws"] >= 2 * c["events"] - P
        assert not np.any(a.output_energies == -99.0)
        assert np.all(np.isfinite(a.output_nus)) and np.all(a.output_nus > 0)
        emitted = a.output_energies >= 0
        assert 0.005 < emitted.mean() < 0.95
        assert np.all(np.abs(a.output_energies) < 10.0 / P)
        assert np.all(a.j_estimator > 0) and np.all(a.nu_bar_estimator > 0)
        assert np.all(a.j_blue_estimator >= 0) and np.all(a.edotlu_estimator >= 0)
        out_nu, out_e = a.output_nus.copy(), a.output_energies.copy()
        jb_sum, ed_sum = a.j_blue_estimator.sum(axis=0), a.edotlu_estimator.sum(axis=0)
        j_a, nubar_a, counters_a = a.j_estimator.copy(), a.nu_bar_estimator.copy(), dict(c)
        # sparse cells of the [L, S] estimators for the second run's comparison (the arrays are 2 x 80 MB)
        rows = np.arange(0, kw["n_lines"], 997)
        jb_rows, ed_rows = a.j_blue_estimator[rows].copy(), a.edotlu_estimator[rows].copy()
        del a
        # a different split into epochs: a smaller log -> more launches, suspended and resumed lanes elsewhere
        eng.set_option("log_capacity", 1_200_000_000)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        assert eng.last_kernel_times()["launches"] > launches
        b = eng.get_results(track_last_interaction=False)
        assert np.array_equal(out_nu, b.output_nus) and np.array_equal(out_e, b.output_energies)
        assert b.counters == counters_a
        assert_allclose(b.j_estimator, j_a, rtol=EST_RTOL)
        assert_allclose(b.nu_bar_estimator, nubar_a, rtol=EST_RTOL)
        assert_allclose(b.j_blue_estimator.sum(axis=0), jb_sum, rtol=EST_RTOL)
        assert_allclose(b.edotlu_estimator.sum(axis=0), ed_sum, rtol=EST_RTOL)
        assert_allclose(b.j_blue_estimator[rows], jb_rows, rtol=EST_RTOL)
        assert_allclose(b.edotlu_estimator[rows], ed_rows, rtol=EST_RTOL)
        del b
        # the CU partition (option pass_cus: propagation and estimator passes on CU-masked streams; measured, never the default): same results
        eng.set_option("pass_cus", 4)
        eng.create_blackbody_packets(P, radius, T_INNER, first=0, count=30_000_000)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        m = eng.get_results(track_last_interaction=False, want_line_estimators=False)
        eng.set_option("pass_cus", 0)
        eng.set_option("log_capacity", 2_500_000_000)
        assert eng.last_kernel_times()["launches"] >= 2  # (2.8e9 records against a log of 1.2e9)
        assert np.array_equal(m.output_nus, out_nu[:30_000_000]) and np.array_equal(m.output_energies, out_e[:30_000_000])
        del m
        # the first 1e5 packets against the oracle
        n = 100_000
        sub, ref = _sample(eng, oracle, prob, P, n, radius)
        assert np.array_equal(out_nu[:n], ref.output_nus) and np.array_equal(out_e[:n], ref.output_energies)
        ha = spectrum.emitted_luminosity_histogram(out_nu[:n], out_e[:n], sub.time_of_simulation, prob.spectrum_frequency_grid)
        hb = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, sub.time_of_simulation, prob.spectrum_frequency_grid)
        assert spectrum.relative_l2(ha, hb) == 0.0  # BASELINE's parity metric (target 1e-6) on the sample
    finally:
        eng.close()


def test_baseline_config5_per_gpu_share_at_its_own_6p25e7_packets(oracle):
    """BASELINE configs[4]: 5e8 packets on 8 GPUs = 6.25e7 per GPU, 100 shells, macroatom, ten v-packets per interaction
    (5e10 v-packets per GPU and iteration), heavy-tailed macro-atom blocks, as `bench.py --config 5` runs it."""
    from tardis_amd.engine import Engine
    kw = dict(synthetic.BASELINE_CONFIGS[5])
    P = int(kw.pop("n_packets")) // 8
    assert P == 62_500_000 and kw["n_shells"] == 100 and kw["n_vpackets"] == 10
    prob = synthetic.make_problem(seed=1, n_packets=1, level_sizes="heavy", **kw)
    radius = float(prob.geometry.r_inner[0])
    eng = Engine(0)
    try:
        eng.set_option("track_last_interaction", 0)
        eng.set_geometry(prob.geometry, prob.time_explosion)
        eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        eng.create_blackbody_packets(P, radius, T_INNER)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        assert eng.last_variant() == 2  # wave kernel with pooled volleys (v-packet screening on)
        a = eng.get_results(track_last_interaction=False, want_line_estimators=False)
        c = a.counters
        assert c["packets"] == P and c["events"] >= P and c["vpackets"] > 100 * P and c["vpacket_line_visits"] > 50 * c["vpackets"]
        assert c["vpackets"] % 10 == 0  # whole volleys
        assert c["rng_draws"] >= c["vpackets"] + c["events"]
        assert not np.any(a.output_energies == -99.0) and np.all(np.isfinite(a.output_nus)) and np.all(a.output_nus > 0)
        assert np.all(a.v_packets_energy_hist >= 0) and a.v_packets_energy_hist.sum() > 0
        assert np.all(a.j_estimator > 0) and np.all(a.nu_bar_estimator > 0)
        n = 2_000
        sub, ref = _sample(eng, oracle, prob, P, n, radius)
        assert np.array_equal(a.output_nus[:n], ref.output_nus) and np.array_equal(a.output_energies[:n], ref.output_energies)
        # the sample on its own through the engine, too: its v-packet spectrum against the oracle's (the full run's histogram
        # holds 3e4 times as many v-packets)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        s = eng.get_results(track_last_interaction=False, want_line_estimators=False)
        assert np.array_equal(s.output_nus, ref.output_nus)
        assert_allclose(s.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=1e-10, atol=1e-300)
        for k in ("vpackets", "vpacket_line_visits", "rng_draws", "events"):
            assert s.counters[k] == ref.counters[k], k
    finally:
        eng.close()
