"""The result-holder half of the drop-in boundary on the GPU: MCTransportSolverHIP.run,
MonteCarloTransportState's properties (montecarlo_transport_state.py:106-160) and tracker_last_interaction_df
(tracker_last_interaction_util.py:33-134), with every value derived from the reference-generated golden of the case; the path a
real run_classic takes (modes/classic/solver.py:223-267): a Python LIST of per-packet tracker objects, and output arrays
that are not contiguous."""
import types

import numpy as np
import pytest
from numpy.testing import assert_allclose

import _golden
from tardis_amd import state as st, transport

pytestmark = pytest.mark.gpu
EST_RTOL = 1e-11


@pytest.mark.parametrize("name", ["macroatom_nv3_log", "downbranch_nv0", "scatter_nv0"])
def test_solver_state_properties_and_tracker_dataframe(name):
    prob, g = _golden.load_case(name)
    cfg = prob.montecarlo_configuration
    mode = {0: "scatter", 1: "downbranch", 2: "macroatom"}[int(cfg.LINE_INTERACTION_TYPE)]
    solver = transport.MCTransportSolverHIP(prob.spectrum_frequency_grid, cfg, line_interaction_type=mode,
                                            enable_full_relativity=bool(cfg.ENABLE_FULL_RELATIVITY), device_id=0)
    ts = solver.initialize_transport_state(prob.packet_collection, prob.geometry, prob.opacity_state, prob.time_explosion,
                                           no_of_virtual_packets=int(cfg.NUMBER_OF_VPACKETS))
    hist = solver.run(ts)
    pc = prob.packet_collection
    t_sim = 1.0 / float(g["in_radiation_field_luminosity"])
    # --- properties other TARDIS code reads
    assert_allclose(ts.output_nu, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(ts.output_energy, g["output_energies"], rtol=1e-13, atol=0)
    assert ts.output_nu is pc.output_nus and ts.output_energy is pc.output_energies  # mutated in place
    assert_allclose(ts.j_estimator, g["j_estimator"], rtol=EST_RTOL)
    assert_allclose(ts.nu_bar_estimator, g["nu_bar_estimator"], rtol=EST_RTOL)
    stride = int(g["line_estimator_stride"])
    assert_allclose(ts.j_blue_estimator[::stride], g["j_blue_estimator"], rtol=EST_RTOL)
    assert_allclose(ts.Edotlu_estimator[::stride], g["edotlu_estimator"], rtol=EST_RTOL)
    assert ts.j_blue_estimator.shape == (len(g["in_line_list_nu"]), len(g["in_r_inner"]))  # the reference's [L, S] layout
    assert ts.time_of_simulation == t_sim
    lum = g["output_energies"] / t_sim
    mask = g["output_energies"] >= 0
    assert_allclose(ts.packet_luminosity, lum, rtol=1e-13)
    assert np.array_equal(ts.emitted_packet_mask, mask)
    assert_allclose(ts.emitted_packet_nu, g["output_nus"][mask], rtol=1e-13)
    assert_allclose(ts.reabsorbed_packet_nu, g["output_nus"][~mask], rtol=1e-13)
    assert_allclose(ts.emitted_packet_luminosity, lum[mask], rtol=1e-13)
    assert_allclose(ts.reabsorbed_packet_luminosity, -lum[~mask], rtol=1e-13)
    assert np.all(ts.reabsorbed_packet_luminosity >= 0)
    assert_allclose(hist, g["v_packets_energy_hist"], rtol=EST_RTOL)
    if "vpacket_nus" in g:
        assert ts.virt_logging
        assert_allclose(ts.virt_packet_nus, g["vpacket_nus"], rtol=1e-13)
        assert_allclose(ts.virt_packet_energies, g["vpacket_energies"], rtol=1e-13)
        assert_allclose(ts.virt_packet_initial_mus, g["vpacket_initial_mus"], rtol=1e-13)
        assert_allclose(ts.virt_packet_initial_rs, g["vpacket_initial_rs"], rtol=1e-13)
    # --- the last-interaction data frame (tracker_last_interaction_util.py:115-132)
    df = ts.tracker_last_interaction_df
    assert list(df.columns) == ["event_id", "last_interaction_type", "status", "radius", "shell_id", "before_nu", "before_mu",
                                "before_energy", "after_nu", "after_mu", "after_energy", "line_absorb_id", "line_emit_id"]
    assert df.index.name == "packet_id" and len(df) == pc.number_of_packets
    assert np.array_equal(df["event_id"].to_numpy(), g["trk_interactions_count"])
    names = {-1: "NO_INTERACTION", 1: "BOUNDARY", 2: "LINE", 4: "ESCATTERING"}
    assert list(df["last_interaction_type"].astype(str)) == [names[int(v)] for v in g["trk_interaction_type"]]
    assert set(df["status"].astype(str)) == {"IN_PROCESS"}
    assert np.array_equal(df["shell_id"].to_numpy(), g["trk_shell_id"])
    assert np.array_equal(df["line_absorb_id"].to_numpy(), g["trk_interaction_line_absorb_id"])
    assert np.array_equal(df["line_emit_id"].to_numpy(), g["trk_interaction_line_emit_id"])
    for col in ("radius", "before_nu", "before_mu", "before_energy", "after_nu", "after_mu", "after_energy"):
        assert_allclose(df[col].to_numpy(), g["trk_" + col], rtol=1e-13, atol=0, equal_nan=True, err_msg=col)


class _Tracker:  # what run_classic passes: one TrackerLastInteraction per packet (tracker_last_interaction.py:8-254)
    def __init__(self):
        for f in st.LastInteractionTrackers.F64_FIELDS:
            setattr(self, f, -1.0)
        for f in st.LastInteractionTrackers.I64_FIELDS:
            setattr(self, f, -1)


def test_list_of_trackers_and_strided_outputs():
    prob, g = _golden.load_case("downbranch_nv0")
    pc = prob.packet_collection
    n = pc.number_of_packets
    nus2, ens2 = np.full(2 * n, -99.0), np.full(2 * n, -99.0)
    like = types.SimpleNamespace(initial_radii=pc.initial_radii, initial_nus=pc.initial_nus, initial_mus=pc.initial_mus,
                                 initial_energies=pc.initial_energies, packet_seeds=pc.packet_seeds,
                                 output_nus=nus2[::2], output_energies=ens2[1::2])  # views with a stride of 16 bytes
    trackers = [_Tracker() for _ in range(n)]
    cfg = prob.montecarlo_configuration
    hist, vt, eb, el = transport.montecarlo_transport_with_vpackets(
        like, prob.geometry, prob.time_explosion, prob.opacity_state, cfg, prob.spectrum_frequency_grid, trackers,
        int(cfg.NUMBER_OF_VPACKETS), False, None)
    assert_allclose(nus2[::2], g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(ens2[1::2], g["output_energies"], rtol=1e-13, atol=0)
    assert np.all(nus2[1::2] == -99.0) and np.all(ens2[::2] == -99.0)  # nothing else was touched
    for f in _golden.TRACKER_I64:
        assert [getattr(t, f) for t in trackers] == list(g["trk_" + f]), f
        assert isinstance(getattr(trackers[0], f), int)
    for f in ("radius", "before_nu", "after_mu", "after_energy"):
        assert_allclose(np.array([getattr(t, f) for t in trackers]), g["trk_" + f], rtol=1e-13, atol=0, equal_nan=True, err_msg=f)
    assert_allclose(eb.mean_intensity_total, g["j_estimator"], rtol=EST_RTOL)
    assert vt.nus.shape == (1,)  # the placeholder collection of get_vpacket_tracker (modes/montecarlo_transport.py:228-235)


def test_scatter_mode_with_the_reference_placeholder_tables():
    """line_interaction_type scatter with the (1, 1) / size-1 macro tables OpacityState.to_numba builds
    (opacities/opacity_state.py:199-209), through the whole boundary."""
    prob, g = _golden.load_case("scatter_nv0")
    op = prob.opacity_state
    ref_like = types.SimpleNamespace(
        electron_density=op.electron_density, line_list_nu=op.line_list_nu, tau_sobolev=op.tau_sobolev,
        transition_probabilities=np.zeros((1, 1)), line2macro_level_upper=np.zeros(1, np.int64),
        macro_block_edge_index=np.zeros(1, np.int64), transition_type=np.zeros(1, np.int64),
        destination_level_id=np.zeros(1, np.int64), transition_line_id=np.zeros(1, np.int64))
    pc = prob.packet_collection
    cfg = prob.montecarlo_configuration
    transport.montecarlo_transport_with_vpackets(pc, prob.geometry, prob.time_explosion, ref_like, cfg,
                                                 prob.spectrum_frequency_grid, None, 0, False, None)
    assert_allclose(pc.output_nus, g["output_nus"], rtol=1e-13, atol=0)
    assert_allclose(pc.output_energies, g["output_energies"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", ["macroatom_nv0", "downbranch_nv0"])
def test_resident_run_with_host_packets_holds_the_same_values(name):
    """MCTransportSolverHIP(resident=True): results stay in HBM until read; every property the non-resident test above checks
    has the golden's value, and the second run on the same opacity object skips the table upload."""
    prob, g = _golden.load_case(name)
    cfg = prob.montecarlo_configuration
    mode = {0: "scatter", 1: "downbranch", 2: "macroatom"}[int(cfg.LINE_INTERACTION_TYPE)]
    solver = transport.MCTransportSolverHIP(prob.spectrum_frequency_grid, cfg, line_interaction_type=mode, device_id=0, resident=True)
    for _ in range(2):
        ts = solver.initialize_transport_state(prob.packet_collection, prob.geometry, prob.opacity_state, prob.time_explosion)
        hist = solver.run(ts)
        assert_allclose(ts.output_nu, g["output_nus"], rtol=1e-13, atol=0)
        assert_allclose(ts.output_energy, g["output_energies"], rtol=1e-13, atol=0)
        assert_allclose(ts.j_estimator, g["j_estimator"], rtol=EST_RTOL)
        assert_allclose(ts.nu_bar_estimator, g["nu_bar_estimator"], rtol=EST_RTOL)
        stride = int(g["line_estimator_stride"])
        assert_allclose(ts.j_blue_estimator[::stride], g["j_blue_estimator"], rtol=EST_RTOL)
        assert_allclose(ts.Edotlu_estimator[::stride], g["edotlu_estimator"], rtol=EST_RTOL)
        assert_allclose(hist, g["v_packets_energy_hist"], rtol=EST_RTOL)
        df = ts.tracker_last_interaction_df
        assert np.array_equal(df["event_id"].to_numpy(), g["trk_interactions_count"])
        assert np.array_equal(df["line_emit_id"].to_numpy(), g["trk_interaction_line_emit_id"])
        assert_allclose(df["after_nu"].to_numpy(), g["trk_after_nu"], rtol=1e-13, atol=0, equal_nan=True)
        # host-side reductions of a host-packet run equal numpy's own
        sp = ts.packet_spectrum(prob.spectrum_frequency_grid)
        lum = g["output_energies"] / ts.time_of_simulation
        assert_allclose(sp["emitted_luminosity"], lum[lum >= 0].sum(), rtol=1e-12)


def test_resident_outer_iterations_with_the_device_packet_source():
    """Two outer iterations the way Simulation.iterate runs them (simulation/base.py:419-490), nothing per-packet crossing PCIe:
    packets drawn on the device (seed_offset = iteration), propagated, spectrum / luminosities and the radiation field reduced
    on the device.  Checked against the non-resident drop-in call on the very packets the device drew."""
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=5, n_packets=1, n_shells=8, n_lines=3000, line_interaction_type="macroatom")
    cfg = prob.montecarlo_configuration
    grid = prob.spectrum_frequency_grid
    geo = prob.geometry
    volume = 4.0 / 3.0 * np.pi * (geo.r_outer**3 - geo.r_inner**3)
    n, t_inner = 30_000, 1.0e4
    solver = transport.MCTransportSolverHIP(grid, cfg, line_interaction_type="macroatom", device_id=0, resident=True)
    for iteration in (0, 1):
        ts = solver.initialize_transport_state(None, geo, prob.opacity_state, prob.time_explosion, n_packets=n,
                                               iteration=iteration, temperature_inner=t_inner)
        solver.run(ts)
        sp = ts.packet_spectrum(grid, luminosity_nu_start=2.0e14, luminosity_nu_end=2.0e15)
        rf = ts.radiation_field(volume, w_epsilon=1e-10)
        pc = ts.packet_collection
        assert pc.number_of_packets == n
        # the same packets through the ordinary drop-in call on a second engine
        host = st.PacketCollection(pc.initial_radii, pc.initial_nus, pc.initial_mus, pc.initial_energies, pc.packet_seeds,
                                   pc.radiation_field_luminosity)
        if iteration == 1:  # iterations draw different packets (seed_offset)
            assert not np.array_equal(pc.packet_seeds, first_seeds)
        first_seeds = pc.packet_seeds
        with Engine(0) as other:
            hist, vt, eb, el = transport.montecarlo_transport_with_vpackets(host, geo, prob.time_explosion, prob.opacity_state, cfg,
                                                                            grid, None, 0, False, None, engine=other)
        assert np.array_equal(ts.output_nu, host.output_nus) and np.array_equal(ts.output_energy, host.output_energies)
        assert_allclose(ts.j_estimator, eb.mean_intensity_total, rtol=EST_RTOL)
        assert_allclose(ts.j_blue_estimator, el.mean_intensity_blueward, rtol=EST_RTOL)
        from tardis_amd import spectrum
        assert_allclose(sp["montecarlo_emitted_luminosity"],
                        spectrum.emitted_luminosity_histogram(host.output_nus, host.output_energies, host.time_of_simulation, grid), rtol=1e-11)
        lum = host.output_energies / host.time_of_simulation
        win = (host.output_nus >= 2.0e14) & (host.output_nus < 2.0e15)
        assert_allclose(sp["emitted_luminosity"], lum[win & (lum >= 0)].sum(), rtol=1e-11)
        assert_allclose(sp["reabsorbed_luminosity"], -lum[win & (lum < 0)].sum(), rtol=1e-11)
        # the radiation field against the oracle of that row on the non-resident estimators
        from oracle import radfield
        t_rad, w, jb = radfield.solve(eb.mean_intensity_total, eb.mean_frequency, el.mean_intensity_blueward.copy(), prob.time_explosion,
                                      host.time_of_simulation, volume, prob.opacity_state.line_list_nu, w_epsilon=1e-10,
                                      detailed_optical_window=False)
        assert_allclose(rf["t_radiative"], t_rad, rtol=1e-10)
        assert_allclose(rf["dilution_factor"], w, rtol=1e-10)
        assert_allclose(rf["j_blues"], jb, rtol=1e-10)
    # a stale view must not hand out another run's data
    ts_old = ts
    ts2 = solver.initialize_transport_state(None, geo, prob.opacity_state, prob.time_explosion, n_packets=100, iteration=2,
                                            temperature_inner=t_inner)
    solver.run(ts2)
    with pytest.raises(RuntimeError):
        ts_old.packet_spectrum(grid)


def test_show_progress_bars_follows_the_running_call(monkeypatch):
    """show_progress_bars (modes/montecarlo_transport.py:94-120: the reference's packet bar advances once per packet): the wrapper
    polls tardis_mc_progress -- packets handed to the propagation kernel so far -- from a thread while the call blocks.  The values
    never decrease, stay within the call's packet count, end AT it, and the call's results are those of a silent call."""
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=3, n_packets=3_000_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    pc = prob.packet_collection
    seen = []
    real = transport._PacketProgress._update

    def spy(self, final=False):
        real(self, final)
        seen.append((self.seen, final))

    monkeypatch.setattr(transport._PacketProgress, "_update", spy)
    with Engine(0) as eng:
        args = (prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration, prob.spectrum_frequency_grid, None, 0)
        transport.montecarlo_transport_with_vpackets(pc, *args, show_progress_bars=False, engine=eng)
        assert not seen
        quiet_nus, quiet_en = pc.output_nus.copy(), pc.output_energies.copy()
        assert eng.progress() == (pc.number_of_packets, pc.number_of_packets)  # a finished call
        pc.output_nus[:] = -99.0
        monkeypatch.setattr(transport._PacketProgress, "INTERVAL", 0.001)  # (the call lasts ~15 ms)
        transport.montecarlo_transport_with_vpackets(pc, *args, show_progress_bars=True, engine=eng)
    values = [v for v, _ in seen]
    assert values and values == sorted(values) and 0 <= values[0] and values[-1] == pc.number_of_packets and seen[-1][1]
    assert np.array_equal(pc.output_nus, quiet_nus) and np.array_equal(pc.output_energies, quiet_en)


# ---- result streaming (round 6; include/tardis_mc.h: tardis_mc_stream_results) ------------------------------------------------------------
def _engine_run(prob, n_track, stream, **options):
    """One call on the engine the way the wrapper makes it; `stream`: the caller's arrays are registered before propagate."""
    from tardis_amd.engine import Engine
    P = prob.packet_collection.initial_nus.size
    out_nu, out_en = np.full(P, -7.0), np.full(P, -7.0)
    trackers = st.LastInteractionTrackers(P) if n_track else None
    with Engine(0) as eng:
        for k, v in options.items():
            eng.set_option(k, v)
        eng.set_option("track_last_interaction", int(n_track))
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators()
        if stream:
            eng.stream_results(out_nu, out_en, trackers)
        eng.propagate(); eng.synchronize()
        res = eng.get_results(out_nu, out_en, track_last_interaction=bool(n_track), trackers=trackers)
        assert res.output_nus is out_nu
        return res, eng.streamed_packets(), eng.last_kernel_times()["launches"]


@pytest.mark.parametrize("track", [True, False], ids=["trackers", "outputs-only"])
@pytest.mark.parametrize("n_packets,capacity", [(1_500_000, 1 << 22), (150_000, 1 << 19)], ids=["ranges", "all-in-flight"])
def test_streamed_results_are_the_results(n_packets, capacity, track):
    """A call that runs as several launches copies the results of the packets handed out so far into the caller's arrays beside the next launch;
    packets that were in flight when their range went out are sent again by get_results.  Same bits in all sixteen arrays as the copy at the end --
    with more packets than the device holds lanes (ranges of finished packets + a late list) and with fewer (everything is in flight at the first
    boundary: the whole result is 'late')."""
    from tardis_amd import synthetic
    prob = synthetic.make_problem(seed=31, n_packets=n_packets, n_shells=20, n_lines=30_000, line_interaction_type="macroatom")
    opts = dict(log_capacity=capacity, stream_min_packets=4096)
    ref, (s0, _), launches0 = _engine_run(prob, track, False, **opts)
    got, (streamed, resent), launches = _engine_run(prob, track, True, **opts)
    assert s0 == 0 and launches0 >= 3 and launches >= 3
    assert streamed >= 4096 and 0 < resent
    if n_packets > 1_000_000:
        assert streamed > 1_000_000  # (resent counts every boundary's in-flight packets: many launches here)
    assert np.array_equal(got.output_nus, ref.output_nus) and np.array_equal(got.output_energies, ref.output_energies)
    assert not np.any(got.output_nus == -7.0)
    if track:
        for f in st.LastInteractionTrackers.I64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f)), f
        for f in st.LastInteractionTrackers.F64_FIELDS:
            assert np.array_equal(getattr(got.trackers, f), getattr(ref.trackers, f), equal_nan=True), f
    assert_allclose(got.j_blue_estimator, ref.j_blue_estimator, rtol=EST_RTOL)
    assert got.counters == ref.counters


def test_streaming_is_for_one_call_and_for_the_registered_arrays():
    """Registered arrays serve the next propagate only; get_results on OTHER arrays after a streamed call copies everything; a late list that overflows
    (capacity forced down through the packet count it is sized from is not reachable here, so: the arming is simply dropped by the next call)."""
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=32, n_packets=600_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
    P = 600_000
    a_nu, a_en = np.full(P, -7.0), np.full(P, -7.0)
    with Engine(0) as eng:
        eng.set_option("log_capacity", 1 << 21); eng.set_option("stream_min_packets", 4096); eng.set_option("track_last_interaction", 0)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.stream_results(a_nu, a_en, None)
        eng.propagate(); eng.synchronize()
        assert eng.streamed_packets()[0] > 0
        other = eng.get_results(track_last_interaction=False)           # fresh arrays: a full copy
        mine = eng.get_results(a_nu, a_en, track_last_interaction=False)  # the registered ones: the remainder + the late list
        assert np.array_equal(other.output_nus, mine.output_nus) and np.array_equal(other.output_energies, mine.output_energies)
        assert not np.any(other.output_nus == -7.0)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()      # not armed any more
        assert eng.streamed_packets() == (0, 0)
        again = eng.get_results(track_last_interaction=False)
        assert np.array_equal(again.output_nus, mine.output_nus)


def test_the_wrapper_streams_into_the_callers_arrays():
    """montecarlo_transport_with_vpackets registers packet_collection.output_* and the trackers itself (in-place outputs of the reference:
    montecarlo_main_loop writes them packet by packet, base.py:168-173)."""
    from tardis_amd import synthetic
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=33, n_packets=800_000, n_shells=20, n_lines=30_000, line_interaction_type="macroatom")
    cfg = prob.montecarlo_configuration
    outs = []
    for min_packets in (4096, 1 << 40):  # streamed | not
        pc = prob.packet_collection
        pc.output_nus[:] = -7.0; pc.output_energies[:] = -7.0
        trk = st.LastInteractionTrackers(pc.initial_nus.size)
        with Engine(0) as eng:
            eng.set_option("log_capacity", 1 << 21); eng.set_option("stream_min_packets", min_packets)
            transport.montecarlo_transport_with_vpackets(pc, prob.geometry, prob.time_explosion, prob.opacity_state, cfg, prob.spectrum_frequency_grid,
                                                         trk, 0, False, None, engine=eng)
            assert (eng.streamed_packets()[0] > 0) == (min_packets == 4096)
        outs.append((pc.output_nus.copy(), pc.output_energies.copy(), trk))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for f in st.LastInteractionTrackers.I64_FIELDS + st.LastInteractionTrackers.F64_FIELDS:
        assert np.array_equal(getattr(outs[0][2], f), getattr(outs[1][2], f), equal_nan=True), f
