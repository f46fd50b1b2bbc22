"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle and the reference-generated
golden vectors.  Bars:
  * integer / index results (status sign, line ids, shell ids, interaction counts): exact;
  * per-packet floating-point results (output_nus, output_energies, tracker fields, v-packet log): BIT-EXACT vs the
    oracle in portable-math mode (same arithmetic on both sides), and rtol 1e-13 vs the reference's own output
    (the reference's regression tolerance, tests/test_montecarlo_main_loop.py);
  * estimators (J, nu_bar, j_blue, Edotlu, v-hist): atomics change the summation order -> rtol 1e-11.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st, synthetic

pytestmark = pytest.mark.gpu

EST_RTOL = 1e-11


@pytest.fixture(scope="module")
def engine():
    from tardis_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


def run_hip(engine, prob, track=True):
    from tardis_amd import transport
    pc = prob.packet_collection
    pc.output_nus[:] = -99.0
    pc.output_energies[:] = -99.0
    trackers = st.LastInteractionTrackers(pc.number_of_packets) if track else None
    hist, vt, eb, el = transport.montecarlo_transport_with_vpackets(
        pc, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
        prob.spectrum_frequency_grid, trackers, prob.montecarlo_configuration.NUMBER_OF_VPACKETS, False, None,
        engine=engine)
    return hist, vt, eb, el, trackers, transport.montecarlo_transport_with_vpackets.last_counters


def run_oracle(oracle, prob, n_threads=1):
    return oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state,
                      prob.montecarlo_configuration, prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE,
                      n_threads=n_threads)


def test_device_arithmetic_is_ieee_and_unfused(engine, oracle):
    rng = np.random.default_rng(7)
    n = 200_000
    x = rng.random(n) * 10.0 ** rng.integers(-20, 20, n)
    y = (rng.random(n) + 1e-3) * 10.0 ** rng.integers(-20, 20, n)
    assert np.array_equal(engine.debug_eval(0, x, y), x + y)
    assert np.array_equal(engine.debug_eval(1, x, y), x * y)
    assert np.array_equal(engine.debug_eval(2, x, y), x / y)
    assert np.array_equal(engine.debug_eval(3, x), np.sqrt(x))
    assert np.array_equal(engine.debug_eval(6, x, y), x * y + x)      # a*b+a must not be contracted to an fma
    assert np.array_equal(engine.debug_eval(8, x - 5.0), np.floor(x - 5.0))
    xi = rng.random(n)
    assert np.array_equal(engine.debug_eval(4, xi), oracle.log_array(xi, 1))
    t = -rng.random(n) * 760.0
    assert np.array_equal(engine.debug_eval(5, t), oracle.exp_array(t, 1))
    # edge arguments of the portable log/exp
    edge = np.array([2.0**-53, 1 - 2.0**-53, 0.5, 0.70710678118654757, 0.7071067811865476, 0.999, 1e-300])
    assert np.array_equal(engine.debug_eval(4, edge), oracle.log_array(edge, 1))
    assert engine.debug_eval(4, np.array([0.0]))[0] == -np.inf


def test_exact_division(engine):
    """The kernels divide with a precomputed reciprocal + two fmas (Markstein); it must equal IEEE division bit for bit."""
    rng = np.random.default_rng(11)
    n = 2_000_000
    def rand(n):
        m = rng.integers(0, 2**52, n, dtype=np.uint64)
        kind = rng.integers(0, 6, n)
        m = np.where(kind == 1, np.uint64(2**52 - 1) - (m & np.uint64(0xff)), m)
        m = np.where(kind == 2, m & np.uint64(0xff), m)
        m = np.where(kind == 3, m & np.uint64(0xfffff00000000), m)
        e = rng.integers(1023 - 100, 1023 + 100, n).astype(np.uint64)
        return ((e << np.uint64(52)) | m).view(np.float64) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    a, b = rand(n), rand(n)
    assert np.array_equal(engine.debug_eval(9, a, b), a / b)
    # the ranges the kernels actually use, plus the guarded extremes (zero numerator, tiny / huge divisors)
    a = np.concatenate([rng.random(1000) * 1e15, [0.0, 1.0, 1e-300, 1e300, 5.0, 3.0]])
    b = np.concatenate([rng.random(1000) * 1e15 + 1.0, [3.0, 1e-310, 1e300, 1e-300, 0.0, np.inf]])
    with np.errstate(divide="ignore"):
        assert np.array_equal(engine.debug_eval(9, a, b), a / b)


@pytest.mark.parametrize("seed", [0, 1, 1963, 23111963, 2**32 - 2])
def test_device_mt19937_matches_numpy_stream(engine, oracle, seed):
    got = engine.debug_eval(7, np.array([float(seed)]), n=1500)
    assert np.array_equal(got, oracle.mt19937_random(seed, 1500))


@pytest.mark.parametrize("name", _golden.CASES)
def test_hip_matches_oracle_and_reference_on_golden_cases(engine, oracle, name):
    _check_golden_case(engine, oracle, name)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("name", [n for n in _golden.CASES if "_nv2" in n or "_nv3" in n or "_nv10" in n])
def test_every_kernel_variant_on_vpacket_golden_cases(engine, oracle, name, variant):
    """Whatever the automatic choice for a v-packet problem is, the group kernel, the wave-owner kernel's pooled volleys, its
    volley queue (variant 4: v-packets traced by a kernel of their own between its launches) and the lane kernel must all
    reproduce the same goldens (incl. the consolidated v-packet log)."""
    engine.set_option("variant", variant)
    try:
        _check_golden_case(engine, oracle, name)
    finally:
        engine.set_option("variant", -1)


def _check_golden_case(engine, oracle, name):
    prob, g = _golden.load_case(name)
    ref = run_oracle(oracle, prob)
    hist, vt, eb, el, trk, counters = run_hip(engine, prob)
    pc = prob.packet_collection
    # per-packet: bit-exact vs oracle, 1e-13 vs the reference
    assert np.array_equal(pc.output_nus, ref.output_nus)
    assert np.array_equal(pc.output_energies, ref.output_energies)
    assert_allclose(pc.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(pc.output_energies, g["output_energies"], rtol=1e-13, atol=0)
    for f in _golden.TRACKER_I64:
        assert np.array_equal(getattr(trk, f), g["trk_" + f]), f
    for f in _golden.TRACKER_F64:
        assert np.array_equal(getattr(trk, f), getattr(ref.trackers, f), equal_nan=True), f
    assert np.all(np.isnan(trk.mu))
    # estimators
    stride = int(g["line_estimator_stride"])
    assert_allclose(eb.mean_intensity_total, ref.j_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(eb.mean_frequency, ref.nu_bar_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(el.mean_intensity_blueward, ref.j_blue_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(el.energy_deposition_line_rate, ref.edotlu_estimator, rtol=EST_RTOL, atol=0)
    assert_allclose(el.mean_intensity_blueward[::stride], g["j_blue_estimator"], rtol=EST_RTOL, atol=0)
    assert_allclose(hist, ref.v_packets_energy_hist, rtol=EST_RTOL, atol=0)
    assert_allclose(hist, g["v_packets_energy_hist"], rtol=EST_RTOL, atol=0)
    if "vpacket_nus" in g:
        assert np.array_equal(vt.nus, ref.vpacket_nus) and np.array_equal(vt.energies, ref.vpacket_energies)
        assert np.array_equal(vt.initial_mus, ref.vpacket_initial_mus) and np.array_equal(vt.initial_rs, ref.vpacket_initial_rs)
        assert_allclose(vt.nus, g["vpacket_nus"], rtol=1e-13, atol=0)
    # work counters are integers: exact
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws", "packets"):
        assert counters[k] == ref.counters[k], k


@pytest.mark.parametrize("mode,n_v,full", [("downbranch", 0, False), ("macroatom", 0, False), ("scatter", 0, False),
                                           ("macroatom", 2, False), ("downbranch", 0, True)])
def test_hip_matches_oracle_on_config1_shape(engine, oracle, mode, n_v, full):
    """BASELINE configs[0] shape (20 shells, 3e4 lines) at 2e4 packets (2e3 with v-packets)."""
    prob = synthetic.make_problem(seed=3, n_packets=2_000 if n_v else 20_000, n_shells=20, n_lines=30_000,
                                  line_interaction_type=mode, n_vpackets=n_v, enable_full_relativity=full)
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    hist, vt, eb, el, trk, counters = run_hip(engine, prob)
    pc = prob.packet_collection
    assert np.array_equal(pc.output_nus, ref.output_nus)
    assert np.array_equal(pc.output_energies, ref.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS:
        assert np.array_equal(getattr(trk, f), getattr(ref.trackers, f)), f
    assert_allclose(eb.mean_intensity_total, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(eb.mean_frequency, ref.nu_bar_estimator, rtol=EST_RTOL)
    assert_allclose(el.mean_intensity_blueward, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(el.energy_deposition_line_rate, ref.edotlu_estimator, rtol=EST_RTOL)
    assert_allclose(hist, ref.v_packets_energy_hist, rtol=EST_RTOL)
    assert counters["line_visits"] == ref.counters["line_visits"] and counters["events"] == ref.counters["events"]
    # the BASELINE parity metric: relative L2 of the real-packet spectrum (target <= 1e-6; here it is exactly 0)
    from tardis_amd import spectrum
    a = spectrum.emitted_luminosity_histogram(pc.output_nus, pc.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    b = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    assert spectrum.relative_l2(a, b) <= 1e-6


def test_edge_cases(engine, oracle):
    # empty packet collection
    prob = synthetic.make_problem(seed=2, n_packets=0, n_shells=4, n_lines=100)
    hist, vt, eb, el, trk, counters = run_hip(engine, prob)
    assert counters["packets"] == 0 and not eb.mean_intensity_total.any() and not el.mean_intensity_blueward.any()
    # one packet, one shell, one line
    prob = synthetic.make_problem(seed=2, n_packets=1, n_shells=1, n_lines=1, line_interaction_type="scatter")
    ref = run_oracle(oracle, prob)
    run_hip(engine, prob)
    assert np.array_equal(prob.packet_collection.output_nus, ref.output_nus)
    # ragged count (not a multiple of the wave or block size) and disabled line scattering with tau = 0
    prob = synthetic.make_problem(seed=5, n_packets=1003, n_shells=7, n_lines=333, line_interaction_type="downbranch",
                                  disable_line_scattering=True)
    prob.opacity_state.tau_sobolev[:] = 0.0    # what OpacitySolver does (opacity_solver.py:46-56)
    ref = run_oracle(oracle, prob)
    run_hip(engine, prob)
    assert np.array_equal(prob.packet_collection.output_nus, ref.output_nus)
    assert np.array_equal(prob.packet_collection.output_energies, ref.output_energies)
    # electron scattering switched off (sigma_T = 1e-200, solver.py:291-300)
    prob = synthetic.make_problem(seed=6, n_packets=500, n_shells=5, n_lines=400, line_interaction_type="macroatom")
    prob.montecarlo_configuration.DISABLE_ELECTRON_SCATTERING = True
    ref = run_oracle(oracle, prob)
    run_hip(engine, prob)
    assert np.array_equal(prob.packet_collection.output_nus, ref.output_nus)


def test_error_codes_map_to_reference_exceptions(engine):
    from tardis_amd.engine import MacroAtomError, MonteCarloException
    # a zig-zag line list (not sorted descending) makes comov_nu - nu_line negative -> MonteCarloException
    # (calculate_distances.py:105-106); the oracle fails on the same packet (first_error_packet == 1)
    prob = synthetic.make_problem(seed=8, n_packets=256, n_shells=3, n_lines=20000, line_interaction_type="scatter")
    prob.opacity_state.line_list_nu[1::2] *= 1.5
    with pytest.raises(MonteCarloException) as ei:
        run_hip(engine, prob)
    assert ei.value.packet_index == 1
    # un-normalised transition probabilities (all zero) -> MacroAtomError
    prob = synthetic.make_problem(seed=9, n_packets=64, n_shells=3, n_lines=200, line_interaction_type="downbranch",
                                  log_tau_mean=1.0)
    prob.opacity_state.transition_probabilities[:] = 0.0
    with pytest.raises(MacroAtomError) as ei:
        run_hip(engine, prob)
    assert ei.value.packet_index == 3


def test_partition_invariance_and_determinism(engine):
    """Size-independent properties used at BASELINE scale: per-packet results do not depend on how packets are
    batched or scheduled; estimators add across batches."""
    prob = synthetic.make_problem(seed=4, n_packets=50_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    hist, vt, eb, el, _, c_all = run_hip(engine, prob, track=False)
    nus, ens = prob.packet_collection.output_nus.copy(), prob.packet_collection.output_energies.copy()
    run_hip(engine, prob, track=False)
    assert np.array_equal(nus, prob.packet_collection.output_nus)          # run-to-run determinism
    J = np.zeros_like(eb.mean_intensity_total)
    jb = np.zeros_like(el.mean_intensity_blueward)
    visits = 0
    full = prob.packet_collection
    for r in range(3):
        prob.packet_collection = full.shard(r, 3)
        _, _, eb_r, el_r, _, c_r = run_hip(engine, prob, track=False)
        J += eb_r.mean_intensity_total
        jb += el_r.mean_intensity_blueward
        visits += c_r["line_visits"]
    prob.packet_collection = full
    assert np.array_equal(full.output_nus, nus) and np.array_equal(full.output_energies, ens)
    assert_allclose(J, eb.mean_intensity_total, rtol=EST_RTOL)
    assert_allclose(jb, el.mean_intensity_blueward, rtol=EST_RTOL)
    assert visits == c_all["line_visits"]
    assert not np.any(full.output_energies == -99.0)


def test_rccl_allreduce_single_rank(engine, oracle):
    """The RCCL binding (dlopen, unique id, communicator, in-place all-reduce of the estimator block) on one rank:
    the reduction over a 1-rank communicator must leave the estimators unchanged."""
    prob = synthetic.make_problem(seed=21, n_packets=20_000, n_shells=10, n_lines=4000, line_interaction_type="macroatom")
    from tardis_amd.engine import Engine
    with Engine(0) as eng:
        with pytest.raises(RuntimeError):
            eng.comm_check()  # (no communicator yet: TARDIS_MC_ERR_STATE)
        eng.comm_init(0, 1, Engine.comm_unique_id())
        assert eng.comm_check() == 1  # the self-check of round 6: the all-reduce of (rank + 1) came back as N (N + 1) / 2, N = 1
        eng.set_geometry(prob.geometry, prob.time_explosion)
        eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        eng.set_packets(prob.packet_collection)
        eng.reset_estimators()
        eng.propagate()
        eng.synchronize()
        before = eng.get_results(track_last_interaction=False)
        eng.allreduce_estimators()
        eng.synchronize()
        after = eng.get_results(track_last_interaction=False)
    assert np.array_equal(before.j_estimator, after.j_estimator)
    assert np.array_equal(before.j_blue_estimator, after.j_blue_estimator)
    assert np.array_equal(before.edotlu_estimator, after.edotlu_estimator)
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    assert np.array_equal(after.output_nus, ref.output_nus)
    assert_allclose(after.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)


def test_baseline_config2_full_size(engine, oracle):
    """BASELINE.json configs[1] at full size (1e7 packets, 20 shells, 3e4 lines, downbranch) through size-independent
    properties: every packet terminates, the work counters add up, a call split into several epochs by a small line-visit
    log (the lanes are suspended and resumed) reproduces the single launch bit for bit per packet and to 1e-11 in the
    estimators, and a 1e5-packet sample equals the oracle."""
    from tardis_amd import spectrum
    prob = synthetic.make_problem(seed=1, **synthetic.BASELINE_CONFIGS[2])
    pc = prob.packet_collection
    P = pc.number_of_packets
    assert P == 10_000_000
    hist, vt, eb, el, _, c1 = run_hip(engine, prob, track=False)
    assert engine.last_kernel_times()["launches"] <= 2  # (one epoch: the log holds the whole call; + the drain, split off into a launch of its own)
    nus, ens = pc.output_nus.copy(), pc.output_energies.copy()
    assert c1["packets"] == P and c1["events"] >= P and c1["line_visits"] >= c1["events"]
    assert not np.any(ens == -99.0) and np.all(np.isfinite(nus)) and np.all(nus > 0)
    emitted = ens >= 0
    assert 0.2 < emitted.mean() < 0.6
    # several epochs
    engine.set_option("log_capacity", 60_000_000)
    _, _, eb2, el2, _, c2 = run_hip(engine, prob, track=False)
    assert engine.last_kernel_times()["launches"] >= 3
    engine.set_option("log_capacity", 2_500_000_000)
    assert np.array_equal(pc.output_nus, nus) and np.array_equal(pc.output_energies, ens)
    assert c1 == c2
    assert_allclose(eb2.mean_intensity_total, eb.mean_intensity_total, rtol=EST_RTOL)
    assert_allclose(el2.mean_intensity_blueward, el.mean_intensity_blueward, rtol=EST_RTOL)
    # oracle on the first 1e5 packets (per-packet results do not depend on batching)
    sub = pc.shard(0, 100)
    ref = oracle.run(sub, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                     prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(),
                     track_last_interaction=False)
    n = sub.number_of_packets
    assert np.array_equal(nus[:n], ref.output_nus) and np.array_equal(ens[:n], ref.output_energies)
    a = spectrum.emitted_luminosity_histogram(nus[:n], ens[:n], pc.time_of_simulation, prob.spectrum_frequency_grid)
    b = spectrum.emitted_luminosity_histogram(ref.output_nus, ref.output_energies, pc.time_of_simulation, prob.spectrum_frequency_grid)
    assert spectrum.relative_l2(a, b) == 0.0


def test_device_packet_spectrum_matches_numpy(engine):
    """real-packet spectrum and filtered luminosities reduced on the device.  Bin ASSIGNMENT follows
    numpy.histogram exactly (checked through exact per-bin sums); the weighted sums agree with numpy's own
    (cumsum-difference) result to its accuracy."""
    import math
    from tardis_amd import spectrum
    prob = synthetic.make_problem(seed=41, n_packets=300_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch",
                                  n_bins=2000)
    run_hip(engine, prob, track=False)
    pc = prob.packet_collection
    grid = prob.spectrum_frequency_grid
    t = pc.time_of_simulation
    lo, hi = grid[200], grid[1500]
    got = engine.packet_spectrum(t, lo, hi)
    # exact reference: numpy's bin assignment (searchsorted rules) + correctly rounded per-bin sums
    for sign, key, lkey in ((1, "montecarlo_emitted_luminosity", "emitted_luminosity"),
                            (-1, "montecarlo_reabsorbed_luminosity", "reabsorbed_luminosity")):
        mask = (pc.output_energies >= 0) if sign > 0 else (pc.output_energies < 0)
        nu, lum = pc.output_nus[mask], sign * (pc.output_energies[mask] / t)
        inside = (nu >= grid[0]) & (nu <= grid[-1])
        idx = np.searchsorted(grid, nu[inside], "right") - 1
        idx[nu[inside] == grid[-1]] = len(grid) - 2
        exact = np.zeros(len(grid) - 1)
        order = np.argsort(idx, kind="stable")
        bounds = np.searchsorted(idx[order], np.arange(len(grid)))
        w = lum[inside][order]
        for b in range(len(grid) - 1):
            exact[b] = math.fsum(w[bounds[b]:bounds[b + 1]])
        assert_allclose(got[key], exact, rtol=1e-12, atol=0)
        assert np.array_equal(got[key] == 0, exact == 0)                       # identical bin occupancy
        np_hist = (spectrum.emitted_luminosity_histogram if sign > 0 else spectrum.reabsorbed_luminosity_histogram)(
            pc.output_nus, pc.output_energies, t, grid)
        assert_allclose(got[key], np_hist, rtol=1e-9, atol=0)
        f = (nu > lo) & (nu < hi)
        assert_allclose(got[lkey], math.fsum(lum[f]), rtol=1e-12)
        assert_allclose(got[lkey], spectrum.calculate_filtered_luminosity(nu, lum, lo, hi), rtol=1e-10)


@pytest.mark.parametrize("options", [
    {"variant": 0}, {"variant": 1}, {"variant": 1, "group_size": 16}, {"variant": 2, "group_size": 4}, {"variant": 2, "group_size": 16},
    {"variant": 2, "log_capacity": 4096},           # line-visit log far too small: most traces take the direct-atomics path
    {"variant": 2, "log_capacity": 0},              # no log at all
    {"variant": 2, "waves_per_simd": 2},
    {"variant": 3}, {"variant": 3, "lane_sweep_min_active": 0}, {"variant": 3, "lane_sweep_min_active": 63, "lane_sweep_max_steps": 1},
    {"variant": 3, "log_capacity": 0},
], ids=lambda o: "-".join(f"{k}{v}" for k, v in o.items()))
@pytest.mark.parametrize("mode", ["downbranch", "macroatom"])
def test_kernel_variants_and_options_agree_with_the_oracle(oracle, options, mode):
    """Every propagation kernel / option combination is the same function of its inputs: per-packet results bit-exact
    against the oracle, estimators within the summation-order tolerance, work counters exact."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=17, n_packets=30_000, n_shells=12, n_lines=9_000, line_interaction_type=mode)
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    eng = Engine(0)
    for k, v in options.items():
        eng.set_option(k, v)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    for _ in range(2):      # the second iteration runs with the log sized from the first one's measured traces per packet
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
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
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 3])
def test_epochs_match_a_single_launch(oracle, variant):
    """A small line-visit log splits a propagate call into epochs (the waves suspend their lanes when their log region is
    full and resume in the next launch, whose estimator passes overlap it on a second stream): same per-packet results,
    trackers and counters as the single launch."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=19, n_packets=(2 << 20) + 777, n_shells=6, n_lines=2_000, line_interaction_type="downbranch")
    outs = []
    for cap in (1_500_000_000, 2_500_000):
        eng = Engine(0)
        eng.set_option("variant", variant)
        eng.set_option("log_capacity", cap)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        outs.append(eng.get_results(track_last_interaction=True))
        launches = eng.last_kernel_times()["launches"]
        assert launches == 1 if cap > 1_000_000_000 else launches >= 3
        eng.close()
    a, b = outs
    assert np.array_equal(a.output_nus, b.output_nus) and np.array_equal(a.output_energies, b.output_energies)
    for f in st.LastInteractionTrackers.I64_FIELDS + st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(a.trackers, f), getattr(b.trackers, f), equal_nan=True), f
    assert_allclose(a.j_blue_estimator, b.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(a.j_estimator, b.j_estimator, rtol=EST_RTOL)
    assert a.counters == b.counters


@pytest.mark.gpu
@pytest.mark.parametrize("level_sizes", ["uniform", "heavy"])
def test_config5_shape_small(oracle, level_sizes):
    """BASELINE configs[4] shape (100 shells, 5e5 lines, macroatom, 10 v-packets) at a packet count the oracle finishes in
    seconds: the largest table sizes of the baseline (24 500 estimator tiles, 144-KiB binning histograms) -- on 4-8-line levels
    and on heavy-tailed macro-atom blocks (what real atomic data looks like; the fixture-sized reference run of this combination
    is the golden macroatom_heavy_100shells_nv10)."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=23, n_packets=3_000, n_shells=100, n_lines=500_000, line_interaction_type="macroatom",
                                  n_vpackets=10, level_sizes=level_sizes)
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    got = eng.get_results(track_last_interaction=True)
    eng.close()
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
    assert_allclose(got.v_packets_energy_hist, ref.v_packets_energy_hist, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k


@pytest.mark.gpu
def test_config5_shape_at_size_properties(oracle):
    """BASELINE configs[4] shape at 3e6 packets (2.4e9 v-packets; its own count is 6.25e7 per GPU): the call the engine routes to the
    wave kernel's pooled volleys with the v-packet screening.  Size-independent properties: every packet terminates, the counters add
    up, the group kernel and the run without the screening reproduce it bit for bit per packet (and the v-packet spectrum to the
    summation order), and the first 2 000 packets equal the oracle."""
    from tardis_amd.engine import Engine
    P = 3_000_000
    prob = synthetic.make_problem(seed=1, n_packets=1, n_shells=100, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=10)
    radius = float(prob.geometry.r_inner[0])
    eng = Engine(0)
    eng.set_option("track_last_interaction", 0)
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.create_blackbody_packets(P, radius, 1.0e4)

    def run(**options):
        for k, v in options.items():
            eng.set_option(k, v)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        r = eng.get_results(track_last_interaction=False, want_line_estimators=False)
        variant = eng.last_variant()
        for k in options:
            eng.set_option(k, -1 if k in ("variant", "vpacket_screening") else 0)
        return r, variant

    a, va = run()
    assert va == 2  # wave kernel, pooled volleys (the screening makes it the faster one from 2.5e6 packets per call on)
    c = a.counters
    assert c["packets"] == P and c["events"] >= P and c["vpackets"] > 100 * P and c["vpacket_line_visits"] > 50 * c["vpackets"]
    assert c["rng_draws"] >= c["vpackets"] + c["events"]
    assert not np.any(a.output_energies == -99.0) and np.all(np.isfinite(a.output_nus)) and np.all(a.output_nus > 0)
    assert np.all(a.v_packets_energy_hist >= 0) and a.v_packets_energy_hist.sum() > 0 and np.all(a.j_estimator > 0)
    b, vb = run(variant=1)              # group kernel, screening on
    d, vd = run(vpacket_screening=0, variant=1)   # line-by-line traces only
    assert vb == 1 and vd == 1
    for other in (b, d):
        assert np.array_equal(a.output_nus, other.output_nus) and np.array_equal(a.output_energies, other.output_energies)
        assert a.counters["vpacket_line_visits"] == other.counters["vpacket_line_visits"] and a.counters["rng_draws"] == other.counters["rng_draws"]
        assert_allclose(other.v_packets_energy_hist, a.v_packets_energy_hist, rtol=1e-10, atol=1e-300)
        assert_allclose(other.j_estimator, a.j_estimator, rtol=EST_RTOL)
    # the first packets against the oracle (per-packet results do not depend on batching)
    n = 2_000
    eng.create_blackbody_packets(P, radius, 1.0e4, first=0, count=n)
    pk = eng.get_packets()
    eng.close()
    sub = st.PacketCollection(pk["initial_radii"], pk["initial_nus"], pk["initial_mus"], pk["initial_energies"], pk["packet_seeds"],
                              4 * np.pi * st.SIGMA_SB * radius**2 * 1.0e4**4)
    ref = oracle.run(sub, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration, prob.spectrum_frequency_grid,
                     math_mode=oracle.MATH_PORTABLE, n_threads=oracle.max_threads(), track_last_interaction=False)
    assert np.array_equal(a.output_nus[:n], ref.output_nus) and np.array_equal(a.output_energies[:n], ref.output_energies)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["downbranch", "macroatom"])
def test_negative_transition_probability_keeps_the_serial_walk(oracle, mode):
    """The wave kernel searches the running sums of the transition probabilities, which presumes they are monotone.  A table
    with a negative entry (the reference just keeps adding) is detected when the sums are built and the problem runs on the
    kernel that walks the block serially: same jumps as the oracle."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=23, n_packets=20_000, n_shells=8, n_lines=4_000, line_interaction_type=mode)
    # in some blocks the first probability becomes negative and the last one takes up the difference: the running sum dips
    # below zero first but still ends at the block's total, so no packet runs out of its block
    tp = prob.opacity_state.transition_probabilities
    edges = np.asarray(prob.opacity_state.macro_block_edge_index)
    n_changed = 0
    for b in range(0, len(edges) - 1, 7):
        lo, hi = int(edges[b]), int(edges[b + 1])
        if hi - lo >= 3:
            first = tp[lo, :].copy()
            tp[lo, :] = -0.1 * first
            tp[hi - 1, :] += 1.1 * first
            n_changed += 1
    assert n_changed > 0
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    got = eng.get_results(track_last_interaction=True)
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k
    eng.close()


@pytest.mark.gpu
def test_log_bounded_epochs_match_the_oracle(oracle):
    """A call whose line-visit log does not fit log_capacity runs as several epochs (suspended and resumed lanes, two log
    buffer sets): same results as the oracle, several launches; repeated, the call reuses its buffers."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=29, n_packets=300_001, n_shells=6, n_lines=2_000, line_interaction_type="downbranch")
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    eng = Engine(0)
    eng.set_option("log_capacity", 1 << 21)      # 4096 waves x 512 records per epoch
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
    for _ in range(2):
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        got = eng.get_results(track_last_interaction=True)
        assert eng.last_kernel_times()["launches"] >= 2
        assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
        assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
        assert_allclose(got.edotlu_estimator, ref.edotlu_estimator, rtol=EST_RTOL)
        assert_allclose(got.j_estimator, ref.j_estimator, rtol=EST_RTOL)
        for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
            assert got.counters[k] == ref.counters[k], k
    eng.close()


@pytest.mark.gpu
def test_negative_optical_depths(oracle):
    """A table with negative Sobolev depths (stimulated emission can produce them): the running optical depth of a trace then
    decreases along the list, which the no-stop bounds of the lane sweeps do not rely on -- the result is the oracle's."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=23, n_packets=20_000, n_shells=10, n_lines=6_000, line_interaction_type="macroatom")
    tau = np.array(prob.opacity_state.tau_sobolev, copy=True)
    tau[::7, ::3] *= -0.25
    prob.opacity_state.tau_sobolev = tau
    ref = run_oracle(oracle, prob, n_threads=oracle.max_threads())
    eng = Engine(0)
    eng.set_option("variant", 3)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.set_packets(prob.packet_collection)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    got = eng.get_results(track_last_interaction=True)
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    for k in ("line_visits", "events", "macro_transitions", "rng_draws"):
        assert got.counters[k] == ref.counters[k], k
    eng.close()
