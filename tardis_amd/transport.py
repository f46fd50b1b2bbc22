"""Drop-in replacement of the reference's Monte Carlo main loop, backed by the HIP engine.

``montecarlo_transport_with_vpackets`` has the signature and return tuple of the reference function
(tardis/transport/montecarlo/modes/montecarlo_transport.py:238-373), so a TARDIS installation switches
engines with one assignment in ``modes/classic/solver.py``.  It accepts the reference's
own jitclass objects or the containers of ``tardis_amd.state`` (duck typing on attribute names).

``MCTransportSolverHIP`` mirrors ``MCTransportSolverClassic.run / run_classic``
(tardis/transport/montecarlo/modes/classic/solver.py:154-273) and ``MonteCarloTransportState`` the result
properties the rest of TARDIS reads (montecarlo_transport_state.py:106-160).
"""
from __future__ import annotations

import os

import numpy as np

from . import state as st
from .engine import Engine, MacroAtomError, MonteCarloException  # noqa: F401  (re-exported)

_engines: dict[int, Engine] = {}


def get_engine(device_id: int | None = None) -> Engine:
    """Process-wide engine per device (one process per GPU: LOCAL_RANK selects the device)."""
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0"))
    eng = _engines.get(device_id)
    if eng is None:
        eng = _engines[device_id] = Engine(device_id)
    return eng


def _fill_trackers(trackers, soa: st.LastInteractionTrackers):
    """Accept the reference's list of per-packet TrackerLastInteraction objects and fill them in place
    (fields of packets/trackers/tracker_last_interaction.py:8-254)."""
    if trackers is None or isinstance(trackers, st.LastInteractionTrackers):
        return
    names = st.LastInteractionTrackers.F64_FIELDS + st.LastInteractionTrackers.I64_FIELDS
    if len(trackers) != len(soa):
        raise ValueError(f"{len(trackers)} trackers for {len(soa)} packets")
    if len(trackers):
        # enable_rpacket_tracking makes run_classic pass TrackerFull objects (array-valued fields, one row per event,
        # packets/trackers/tracker_full.py): the engine records the last interaction only
        first = trackers[0]
        if type(first).__name__ == "TrackerFull" or any(np.ndim(getattr(first, n, 0.0)) != 0 for n in names):
            raise NotImplementedError("full r-packet tracking (TrackerFull, montecarlo.tracking.track_rpacket) is not "
                                      "implemented by the HIP engine; only TrackerLastInteraction is")
    cols = {n: getattr(soa, n).tolist() for n in names}
    for i, t in enumerate(trackers):
        for n in names:
            setattr(t, n, cols[n][i])


class _PacketProgress:
    """The reference's packet progress bar (progress_bars.update_packets_pbar, fed once per packet by the main loop,
    modes/montecarlo_transport.py:94-120) for a call that blocks inside the library: a thread polls Engine.progress() -- packets handed
    to the propagation kernel so far -- and moves a tqdm bar (a plain stderr line without tqdm).  `show_progress_bars=False`: nothing."""

    INTERVAL = 0.2  # seconds between two polls

    def __init__(self, engine, enabled: bool):
        import os
        # (one bar per job: in a multi-process run only rank 0 draws it -- every rank propagates its own shard at the same pace)
        enabled = bool(enabled) and os.environ.get("RANK", "0") in ("0", "")
        self.engine, self.enabled, self.interval = engine, enabled, self.INTERVAL
        self.thread = self.stop = self.bar = None
        self.seen = 0

    def _update(self, final=False):
        try:
            started, total = self.engine.progress()
        except Exception:  # noqa: BLE001 -- a progress bar never takes the run down
            return
        if final:
            started = total
        if self.bar is None:
            try:
                from tqdm.auto import tqdm
                self.bar = tqdm(total=total, desc="Packets", unit="pkt", leave=False)
            except Exception:  # noqa: BLE001
                self.bar = False
        if self.bar:
            self.bar.total = total
            self.bar.update(max(started - self.seen, 0))
        elif started != self.seen or final:
            import sys
            print(f"\rPackets {started}/{total}", end="\n" if final else "", file=sys.stderr, flush=True)
        self.seen = max(self.seen, started)

    def __enter__(self):
        if self.enabled:
            import threading
            self.stop = threading.Event()
            if self.bar and self.seen:  # a second attempt of the same call (the v-packet log was resized): the same bar, from the start
                self.bar.reset()
            self.seen = 0

            def poll():
                while not self.stop.wait(self.interval):
                    self._update()
            self.thread = threading.Thread(target=poll, name="tardis-amd-progress", daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.stop.set()
            self.thread.join()
            if exc[0] is None:
                self._update(final=True)
        return False

    def close(self):
        if self.bar:
            self.bar.close()
        self.bar = None


def montecarlo_transport_with_vpackets(packet_collection, geometry_state_numba, time_explosion: float,
                                       opacity_state_numba, montecarlo_configuration, spectrum_frequency_grid,
                                       trackers, number_of_vpackets: int, show_progress_bars: bool = False,
                                       packet_propagation_function=None, *, engine: Engine | None = None):
    """Run the classic (line + electron scattering) Monte Carlo transport on the GPU.

    Returns ``(v_packets_energy_hist, vpacket_tracker, estimators_bulk, estimators_line)`` and mutates
    ``packet_collection.output_nus / output_energies`` and ``trackers`` in place, exactly like the reference.
    ``packet_propagation_function`` is accepted for signature compatibility; only the classic homologous mode
    is implemented by the engine (anything else must stay on the reference path).
    """
    if packet_propagation_function is not None:
        name = getattr(packet_propagation_function, "__name__", "")
        mod = getattr(packet_propagation_function, "__module__", "") or ""
        if name != "packet_propagation" or "nonhomologous" in mod or "iip" in mod:
            raise NotImplementedError("the HIP engine implements the classic homologous packet_propagation only")
    eng = engine or get_engine()
    eng.set_geometry(geometry_state_numba, time_explosion)
    eng.set_opacity(opacity_state_numba)
    eng.set_config(montecarlo_configuration, spectrum_frequency_grid, number_of_vpackets)
    track = trackers is not None
    eng.set_option("track_last_interaction", int(track))
    eng.set_packets(packet_collection)
    out_nus, out_en = packet_collection.output_nus, packet_collection.output_energies
    in_place = all(isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous for a in (out_nus, out_en))
    vlog_capacity = None
    progress = _PacketProgress(eng, show_progress_bars)  # (one bar per call, whatever the number of attempts)
    try:
        for _attempt in range(2):
            eng.reset_estimators()
            if in_place and _attempt == 0:  # (the caller's arrays are filled while the call runs, launch by launch)
                eng.stream_results(out_nus, out_en, trackers if isinstance(trackers, st.LastInteractionTrackers) and len(trackers) == eng.n_packets else None)
            with progress:
                eng.propagate()
                eng.synchronize()
            res = eng.get_results(out_nus if in_place else None, out_en if in_place else None, track_last_interaction=track,
                                  vpacket_log_capacity=vlog_capacity, trackers=trackers)
            if res.vpacket_log_count <= len(res.vpacket_nus):
                break
            # The v-packet log was sized from a guess and overflowed (the device then drops entries): the run is
            # deterministic, so repeat it with the capacity it asked for -- the reference returns every v-packet.
            vlog_capacity = res.vpacket_log_count
            eng.set_option("vpacket_log_capacity", vlog_capacity)
        else:
            raise RuntimeError("v-packet log overflow persisted after resizing")
    finally:
        progress.close()
        if vlog_capacity is not None:
            eng.set_option("vpacket_log_capacity", 0)  # back to automatic sizing, whichever way the call ended
    if not in_place:
        packet_collection.output_nus[:] = res.output_nus
        packet_collection.output_energies[:] = res.output_energies
    if track:
        if res.trackers is trackers:
            pass  # (the library wrote into the caller's arrays)
        elif isinstance(trackers, st.LastInteractionTrackers):
            for n in st.LastInteractionTrackers.F64_FIELDS + st.LastInteractionTrackers.I64_FIELDS:
                getattr(trackers, n)[:] = getattr(res.trackers, n)
        else:
            _fill_trackers(trackers, res.trackers)
    estimators_bulk = st.EstimatorsBulk(res.j_estimator, res.nu_bar_estimator)
    estimators_line = st.EstimatorsLine(res.j_blue_estimator, res.edotlu_estimator)
    cfg = montecarlo_configuration
    if cfg.ENABLE_VPACKET_TRACKING and number_of_vpackets > 0:
        n = min(res.vpacket_log_count, len(res.vpacket_nus))
        vt = st.VPacketCollection(-1, spectrum_frequency_grid, cfg.VPACKET_SPAWN_START_FREQUENCY,
                                  cfg.VPACKET_SPAWN_END_FREQUENCY, -1, n)
        vt.nus[:] = res.vpacket_nus[:n]
        vt.energies[:] = res.vpacket_energies[:n]
        vt.initial_mus[:] = res.vpacket_initial_mus[:n]
        vt.initial_rs[:] = res.vpacket_initial_rs[:n]
    else:  # placeholder, as get_vpacket_tracker does (modes/montecarlo_transport.py:228-235)
        vt = st.VPacketCollection(-1, spectrum_frequency_grid, cfg.VPACKET_SPAWN_START_FREQUENCY,
                                  cfg.VPACKET_SPAWN_END_FREQUENCY, -1, 1)
    montecarlo_transport_with_vpackets.last_counters = res.counters
    montecarlo_transport_with_vpackets.last_kernel_ms = eng.last_propagate_ms()
    return res.v_packets_energy_hist, vt, estimators_bulk, estimators_line


class DevicePacketCollection:
    """A PacketCollection whose arrays live in the engine: the packets were drawn by the device packet source (SURVEY 8f-1,
    BlackBodySimpleSource.create_packets, packet_source/base.py:195-253) and propagated in place.  Same attribute names as
    the reference's PacketCollection (packets/packet_collections.py:13-101); a host copy of an array is made the first time it
    is read.  The engine's resident packets must still be these (a later set_packets / create_blackbody_packets on the same
    engine invalidates the un-read arrays: reading them then raises instead of returning another run's data)."""

    def __init__(self, engine: Engine, n_packets: int, radius: float, temperature: float, base_seed: int, seed_offset: int):
        self._eng = engine
        self._n = int(n_packets)
        self.radius, self.temperature, self.base_seed, self.seed_offset = float(radius), float(temperature), int(base_seed), int(seed_offset)
        self.radiation_field_luminosity = 4 * np.pi * st.SIGMA_SB * self.radius**2 * self.temperature**4  # base.py:255-270
        self.time_of_simulation = 1 / self.radiation_field_luminosity
        self._inputs = None
        self._outputs = None
        self._pgen = self._rgen = None

    def _draw(self):
        self._eng.create_blackbody_packets(self._n, self.radius, self.temperature, base_seed=self.base_seed, seed_offset=self.seed_offset)
        self._pgen = self._eng.packets_generation
        self._inputs = self._outputs = None

    def _mark_propagated(self):
        self._rgen = self._eng.results_generation
        self._outputs = None

    def _fresh(self, results: bool):
        if self._pgen != self._eng.packets_generation or (results and self._rgen != self._eng.results_generation):
            raise RuntimeError("the engine's resident packets / results are no longer this collection's (it ran something else since)")

    def _in(self, name):
        if self._inputs is None:
            self._fresh(False)
            self._inputs = self._eng.get_packets()
        return self._inputs[name]

    def _out(self, k):
        if self._outputs is None:
            self._fresh(True)
            r = self._eng.get_results(track_last_interaction=False, want_line_estimators=False)
            self._outputs = (r.output_nus, r.output_energies)
        return self._outputs[k]

    number_of_packets = property(lambda self: self._n)
    initial_radii = property(lambda self: self._in("initial_radii"))
    initial_nus = property(lambda self: self._in("initial_nus"))
    initial_mus = property(lambda self: self._in("initial_mus"))
    initial_energies = property(lambda self: self._in("initial_energies"))
    packet_seeds = property(lambda self: self._in("packet_seeds"))
    output_nus = property(lambda self: self._out(0))
    output_energies = property(lambda self: self._out(1))


class _DeviceEstimators:
    """EstimatorsLine / the last-interaction trackers of a resident run: fetched from the engine on first access."""

    def __init__(self, engine: Engine):
        self._eng, self._rgen, self._egen, self._line, self._trk = engine, engine.results_generation, engine.estimators_generation, None, None

    def rebase(self):
        """The owner of the run changed the resident estimators on purpose (the N-GPU all-reduce of an outer iteration):
        they are still this run's."""
        self._fresh(estimators=False)
        self._egen = self._eng.estimators_generation

    def _fresh(self, estimators=True):
        if self._rgen != self._eng.results_generation:
            raise RuntimeError("the engine has propagated again since: these estimators are gone")
        if estimators and self._egen != self._eng.estimators_generation:
            raise RuntimeError("the engine's estimators were reset / all-reduced since this run: they are no longer this run's "
                               "(call transport_state.estimators_allreduced() after an intended all-reduce)")

    def _lines(self):
        if self._line is None:
            self._fresh()
            r = self._eng.get_results(track_last_interaction=False, want_packet_outputs=False)
            self._line = st.EstimatorsLine(r.j_blue_estimator, r.edotlu_estimator)
        return self._line

    mean_intensity_blueward = property(lambda self: self._lines().mean_intensity_blueward)
    energy_deposition_line_rate = property(lambda self: self._lines().energy_deposition_line_rate)

    def trackers(self):
        if self._trk is None:
            self._fresh(estimators=False)
            self._trk = self._eng.get_results(track_last_interaction=True, want_line_estimators=False, want_packet_outputs=False).trackers
        return self._trk


class MonteCarloTransportState:
    """Result holder with the reference's property names (montecarlo_transport_state.py:15-317), unit-less."""

    def __init__(self, packet_collection, geometry_state_numba, opacity_state_numba, time_explosion):
        self.packet_collection = packet_collection
        self.geometry_state_numba = geometry_state_numba
        self.opacity_state_numba = opacity_state_numba
        self.time_explosion = time_explosion
        self.estimators_bulk = None
        self.estimators_line = None
        self.vpacket_tracker = None
        self._tracker_last_interaction = None
        self.tracker_full_df = None
        self.enable_full_relativity = False
        self.virt_logging = False
        self._engine = None           # set by a resident run on device packets: the spectrum is then reduced on the device
        self._est_engine = None       # set by every resident run: the engine that holds this run's estimators
        self._device_estimators = None

    @property
    def tracker_last_interaction(self):
        if self._tracker_last_interaction is None and self._device_estimators is not None:
            self._tracker_last_interaction = self._device_estimators.trackers()
        return self._tracker_last_interaction

    @tracker_last_interaction.setter
    def tracker_last_interaction(self, value):
        self._tracker_last_interaction = value

    def packet_spectrum(self, spectrum_frequency_grid, luminosity_nu_start=0.0, luminosity_nu_end=float("inf")):
        """montecarlo_emitted / reabsorbed_luminosity histograms over the spectrum grid and the filtered luminosity sums
        (spectrum/base.py:140-159, spectrum/luminosity.py:5-30): on the device after a resident run, on the host otherwise."""
        if self._engine is not None:
            self.packet_collection._fresh(True)
            return self._engine.packet_spectrum(self.time_of_simulation, luminosity_nu_start, luminosity_nu_end)
        from . import spectrum
        nus, en, t = self.output_nu, self.output_energy, self.time_of_simulation
        win = (nus >= luminosity_nu_start) & (nus < luminosity_nu_end)
        lum = en / t
        return {"montecarlo_emitted_luminosity": spectrum.emitted_luminosity_histogram(nus, en, t, spectrum_frequency_grid),
                "montecarlo_reabsorbed_luminosity": spectrum.reabsorbed_luminosity_histogram(nus, en, t, spectrum_frequency_grid),
                "emitted_luminosity": float(np.sum(lum[win & (en >= 0)])), "reabsorbed_luminosity": float(-np.sum(lum[win & (en < 0)]))}

    def radiation_field(self, volume, w_epsilon=1e-10, detailed_optical_window=False, want_j_blues=True):
        """MCRadiationFieldPropertiesSolver.solve (estimators/mc_rad_field_solver.py:37-144) on the engine's resident (after
        an N-GPU step: all-reduced) estimators.  Only after a resident run."""
        if self._est_engine is None:
            raise RuntimeError("radiation_field() needs a resident run (MCTransportSolverHIP(..., resident=True))")
        self._device_estimators._fresh()
        return self._est_engine.radiation_field(self.time_of_simulation, volume, w_epsilon, detailed_optical_window, want_j_blues)

    def estimators_allreduced(self):
        """Tell the lazy views that the engine's estimators were all-reduced on purpose after this run (N-GPU outer iteration:
        Engine.allreduce_estimators, then radiation_field())."""
        if self._device_estimators is not None:
            self._device_estimators.rebase()

    output_nu = property(lambda self: self.packet_collection.output_nus)
    output_energy = property(lambda self: self.packet_collection.output_energies)
    nu_bar_estimator = property(lambda self: self.estimators_bulk.mean_frequency)
    j_estimator = property(lambda self: self.estimators_bulk.mean_intensity_total)
    j_blue_estimator = property(lambda self: self.estimators_line.mean_intensity_blueward)
    Edotlu_estimator = property(lambda self: self.estimators_line.energy_deposition_line_rate)
    time_of_simulation = property(lambda self: self.packet_collection.time_of_simulation)
    packet_luminosity = property(
        lambda self: self.packet_collection.output_energies / self.packet_collection.time_of_simulation)
    emitted_packet_mask = property(lambda self: self.packet_collection.output_energies >= 0)
    emitted_packet_nu = property(lambda self: self.packet_collection.output_nus[self.emitted_packet_mask])
    reabsorbed_packet_nu = property(lambda self: self.packet_collection.output_nus[~self.emitted_packet_mask])
    emitted_packet_luminosity = property(lambda self: self.packet_luminosity[self.emitted_packet_mask])
    reabsorbed_packet_luminosity = property(lambda self: -self.packet_luminosity[~self.emitted_packet_mask])
    virt_packet_nus = property(lambda self: self.vpacket_tracker.nus)
    virt_packet_energies = property(lambda self: self.vpacket_tracker.energies)
    virt_packet_initial_mus = property(lambda self: self.vpacket_tracker.initial_mus)
    virt_packet_initial_rs = property(lambda self: self.vpacket_tracker.initial_rs)

    @property
    def tracker_last_interaction_df(self):
        """Same columns as trackers_last_interaction_to_df (tracker_last_interaction_util.py:33-134)."""
        import pandas as pd

        t = self.tracker_last_interaction
        names = {-1: "NO_INTERACTION", 1: "BOUNDARY", 2: "LINE", 4: "ESCATTERING", 8: "CONTINUUM_PROCESS"}
        it_dtype = pd.CategoricalDtype(categories=["NO_INTERACTION", "BOUNDARY", "LINE", "ESCATTERING", "CONTINUUM_PROCESS"])
        st_dtype = pd.CategoricalDtype(categories=["IN_PROCESS", "EMITTED", "REABSORBED", "ADIABATIC_COOLING"])
        n = len(t)
        return pd.DataFrame(
            {
                "event_id": t.interactions_count,
                "last_interaction_type": pd.Categorical([names[int(v)] for v in t.interaction_type], dtype=it_dtype),
                "status": pd.Categorical(["IN_PROCESS"] * n, dtype=st_dtype),
                "radius": t.radius, "shell_id": t.shell_id, "before_nu": t.before_nu, "before_mu": t.before_mu,
                "before_energy": t.before_energy, "after_nu": t.after_nu, "after_mu": t.after_mu,
                "after_energy": t.after_energy,
                "line_absorb_id": pd.array(t.interaction_line_absorb_id, dtype="int64"),
                "line_emit_id": pd.array(t.interaction_line_emit_id, dtype="int64"),
            },
            index=pd.RangeIndex(n, name="packet_id"))


class MCTransportSolverHIP:
    """GPU counterpart of MCTransportSolverClassic (modes/classic/solver.py:46-273), plain-array inputs.

    ``resident=True`` is the outer-iteration form (Simulation.iterate, simulation/base.py:419-490: create packets -> run ->
    radiation field + luminosities -> plasma): everything the plasma step does not need stays in HBM.  Packets come from the
    device packet source when ``initialize_transport_state`` is given ``n_packets`` instead of a packet collection; the
    opacity tables are re-uploaded only when the opacity object changed (TARDIS builds a new one per iteration; pass
    ``reuse_opacity=False`` if yours is mutated in place); per-packet outputs, trackers and the [L,S] line estimators are
    fetched on first access of the respective ``transport_state`` attribute; ``transport_state.packet_spectrum`` and
    ``.radiation_field`` reduce on the device.  Values are those of the non-resident path (tests/test_boundary_gpu.py)."""

    def __init__(self, spectrum_frequency_grid, montecarlo_configuration=None, line_interaction_type="macroatom",
                 enable_full_relativity=False, device_id=None, nthreads=1, resident=False, reuse_opacity=True,
                 enable_last_interaction_tracking=True, engine: Engine | None = None):
        self.spectrum_frequency_grid = np.ascontiguousarray(spectrum_frequency_grid, dtype=np.float64)
        self.montecarlo_configuration = montecarlo_configuration or st.MonteCarloConfiguration()
        self.line_interaction_type = line_interaction_type
        self.enable_full_relativity = enable_full_relativity
        self.nthreads = nthreads  # accepted for API compatibility; the GPU engine ignores it
        self.device_id = device_id
        self.resident = bool(resident)
        self.reuse_opacity = bool(reuse_opacity)
        self.enable_last_interaction_tracking = bool(enable_last_interaction_tracking)
        self.transport_state = None
        self._engine = engine          # (default: the process-wide engine of the device)

    def _eng(self) -> Engine:
        return self._engine if self._engine is not None else get_engine(self.device_id)

    def initialize_transport_state(self, packet_collection, geometry, opacity_state, time_explosion,
                                   no_of_virtual_packets=0, *, n_packets=None, iteration=0, temperature_inner=None,
                                   base_seed=None):
        cfg = self.montecarlo_configuration
        cfg.LINE_INTERACTION_TYPE = st.LINE_INTERACTION_TYPES[self.line_interaction_type]
        cfg.NUMBER_OF_VPACKETS = no_of_virtual_packets
        cfg.TEMPORARY_V_PACKET_BINS = no_of_virtual_packets
        cfg.ENABLE_FULL_RELATIVITY = self.enable_full_relativity
        if packet_collection is None:
            # device packet source: BlackBodySimpleSource(radius = r_inner[0], temperature = T_inner, base_seed).create_packets(
            # n_packets, seed_offset = iteration) -- simulation/base.py:419-433
            if not self.resident or n_packets is None or temperature_inner is None:
                raise ValueError("the device packet source needs resident=True, n_packets and temperature_inner")
            seed = cfg.MONTECARLO_SEED if base_seed is None else base_seed
            packet_collection = DevicePacketCollection(self._eng(), n_packets, float(np.asarray(geometry.r_inner)[0]),
                                                       temperature_inner, seed, iteration)
        ts = MonteCarloTransportState(packet_collection, geometry, opacity_state, time_explosion)
        ts.enable_full_relativity = cfg.ENABLE_FULL_RELATIVITY
        return ts

    def run(self, transport_state, show_progress_bars=False):
        return self.run_classic(transport_state, show_progress_bars)

    def run_classic(self, transport_state, show_progress_bars=False):
        if self.resident:
            return self._run_resident(transport_state, show_progress_bars)
        self.transport_state = transport_state
        cfg = self.montecarlo_configuration
        n = len(transport_state.packet_collection.initial_nus)
        trackers = st.LastInteractionTrackers(n)
        hist, vtracker, est_bulk, est_line = montecarlo_transport_with_vpackets(
            transport_state.packet_collection, transport_state.geometry_state_numba, float(transport_state.time_explosion),
            transport_state.opacity_state_numba, cfg, self.spectrum_frequency_grid, trackers, cfg.NUMBER_OF_VPACKETS,
            show_progress_bars, None, engine=self._eng())
        transport_state.estimators_bulk = est_bulk
        transport_state.estimators_line = est_line
        if cfg.ENABLE_VPACKET_TRACKING and cfg.NUMBER_OF_VPACKETS > 0:
            transport_state.vpacket_tracker = vtracker
        transport_state.tracker_last_interaction = trackers
        transport_state.virt_logging = cfg.ENABLE_VPACKET_TRACKING
        return hist

    def _run_resident(self, ts, show_progress_bars=False):
        self.transport_state = ts
        cfg = self.montecarlo_configuration
        if cfg.ENABLE_VPACKET_TRACKING and cfg.NUMBER_OF_VPACKETS > 0:
            raise NotImplementedError("the consolidated v-packet log is a per-v-packet host result: use resident=False with it")
        eng = self._eng()
        eng.set_geometry(ts.geometry_state_numba, float(ts.time_explosion))
        op = ts.opacity_state_numba
        # (residency is tracked on the engine: the default engine is process-wide, and another solver or the non-resident
        # entry point may have uploaded different tables since this solver's last run)
        if not (self.reuse_opacity and eng.resident_opacity is op):
            eng.set_opacity(op)
        eng.set_config(cfg, self.spectrum_frequency_grid, cfg.NUMBER_OF_VPACKETS)
        eng.set_option("track_last_interaction", int(self.enable_last_interaction_tracking))
        pc = ts.packet_collection
        device_packets = isinstance(pc, DevicePacketCollection)
        if device_packets:
            pc._draw()
        else:
            eng.set_packets(pc)
        eng.reset_estimators()
        progress = _PacketProgress(eng, show_progress_bars)
        try:
            with progress:
                eng.propagate()
                eng.synchronize()
        finally:
            progress.close()
        res = eng.get_results(track_last_interaction=False, want_line_estimators=False, want_packet_outputs=False)  # small arrays + error check
        if device_packets:
            pc._mark_propagated()
        else:  # host packets: the reference's in-place outputs
            r = eng.get_results(pc.output_nus, pc.output_energies, track_last_interaction=False, want_line_estimators=False)
            if r.output_nus is not pc.output_nus:
                pc.output_nus[:] = r.output_nus; pc.output_energies[:] = r.output_energies
        ts._engine = eng if device_packets else None  # (host packets: the spectrum comes from the host outputs)
        ts._est_engine = eng
        ts._device_estimators = _DeviceEstimators(eng)
        ts.estimators_bulk = st.EstimatorsBulk(res.j_estimator, res.nu_bar_estimator)
        ts.estimators_line = ts._device_estimators
        ts._tracker_last_interaction = None if self.enable_last_interaction_tracking else st.LastInteractionTrackers(0)
        ts.virt_logging = False
        montecarlo_transport_with_vpackets.last_counters = res.counters
        montecarlo_transport_with_vpackets.last_kernel_ms = eng.last_propagate_ms()
        return res.v_packets_energy_hist
