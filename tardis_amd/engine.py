"""Thin object wrapper over the C ABI (one Engine = one TardisMcContext = one GPU + one stream)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi, _lib
from . import state as st


class MonteCarloException(ValueError):
    """Mirror of tardis.transport.montecarlo.utils.MonteCarloException (utils.py:9)."""


class MacroAtomError(ValueError):
    """Mirror of tardis.transport.montecarlo.macro_atom.MacroAtomError (macro_atom.py:15)."""


class Engine:
    def __init__(self, device_id: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.tardis_mc_create(int(device_id), C.byref(h))
        if rc != 0:
            msg = self._L.tardis_mc_last_error(None).decode()
            raise _lib.EngineUnavailable(f"tardis_mc_create(device {device_id}) failed ({rc}): {msg}")
        self._h = h
        self._keep = {}
        self.n_packets = self.n_shells = self.n_lines = self.n_grid = 0
        self._vpk_log = False
        self._n_v = 0
        self.packets_generation = 0  # bumped whenever the resident packets are replaced (lazy host views check it)
        self.results_generation = 0  # bumped by every propagate() that succeeded
        # bumped by everything that changes the resident estimators: propagate, reset_estimators, allreduce_estimators
        # (lazy views of a resident run check it: a later reset / all-reduce must not be read as that run's estimators)
        self.estimators_generation = 0
        # the opacity object whose tables are in HBM (None: unknown) -- residency is a property of the ENGINE, which several
        # solvers and the non-resident entry point may share (MCTransportSolverHIP reuses the upload only against this)
        self.resident_opacity = None
        self.options = {}  # options set through set_option (name -> last value accepted by the library)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.tardis_mc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what, packet_index=-1):
        if rc == 0:
            return
        msg = self._L.tardis_mc_last_error(self._h).decode()
        exc = None
        if rc == _abi.ERR_MONTECARLO:
            exc = MonteCarloException("nu difference is less than 0.0")
        elif rc == _abi.ERR_MACRO_ATOM:
            exc = MacroAtomError("MacroAtom ran out of the block. This should not happen as the sum of "
                                 "probabilities is normalized to 1 and the probability_event should be less than 1")
        elif rc == _abi.ERR_UNSUPPORTED:
            exc = NotImplementedError("macro-atom transition type outside classic mode (continuum processes)")
        if exc is not None:
            exc.packet_index = packet_index  # lowest failing packet index (the reference aborts on the first one)
            raise exc
        raise RuntimeError(f"{what} failed ({rc}): {msg}")

    # -- staged API
    def set_option(self, name: str, value: int):
        self._check(self._L.tardis_mc_set_option(self._h, name.encode(), int(value)), f"set_option({name})")
        self.options[name] = int(value)

    def set_geometry(self, geometry, time_explosion=None):
        m = _abi.marshal_geometry(geometry, time_explosion)
        self._check(self._L.tardis_mc_set_geometry(self._h, m.ref()), "set_geometry")
        self.n_shells = int(m.struct.n_shells)

    def set_opacity(self, opacity_state):
        m = _abi.marshal_opacity(opacity_state)
        self.resident_opacity = None  # (a failed upload leaves the tables undefined)
        self._check(self._L.tardis_mc_set_opacity(self._h, m.ref()), "set_opacity")
        self.n_lines, self.n_shells = int(m.struct.n_lines), int(m.struct.n_shells)
        self.resident_opacity = opacity_state

    def set_config(self, montecarlo_configuration, spectrum_frequency_grid, number_of_vpackets=None, sigma_thomson=None):
        m = _abi.marshal_config(montecarlo_configuration, spectrum_frequency_grid, number_of_vpackets, sigma_thomson)
        self._check(self._L.tardis_mc_set_config(self._h, m.ref()), "set_config")
        self.n_grid = int(m.struct.n_spectrum_grid)
        self._n_v = int(m.struct.number_of_vpackets)
        self._vpk_log = bool(m.struct.enable_vpacket_tracking) and self._n_v > 0

    def set_packets(self, packet_collection):
        m = _abi.marshal_packets(packet_collection)
        self._check(self._L.tardis_mc_set_packets(self._h, m.ref()), "set_packets")
        self.n_packets = int(m.struct.n_packets)
        self.packets_generation += 1

    def reset_estimators(self):
        self.estimators_generation += 1
        self._check(self._L.tardis_mc_reset_estimators(self._h), "reset_estimators")

    def propagate(self):
        # (whatever happens, the previous run's results are gone; the counters a view compares with only become valid --
        # equal to the view's -- for views created AFTER a successful call)
        self.results_generation += 1
        self.estimators_generation += 1
        self._check(self._L.tardis_mc_propagate(self._h), "propagate")
        self.results_generation += 1
        self.estimators_generation += 1

    def synchronize(self):
        self._check(self._L.tardis_mc_synchronize(self._h), "synchronize")

    def progress(self) -> tuple[int, int]:
        """(packets handed to the propagation kernel so far, packets of the call) of the running -- or the last -- propagate().
        Meant to be polled from another thread while propagate() blocks (tardis_mc_progress reads one device word on its own stream)."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._L.tardis_mc_progress(self._h, C.byref(a), C.byref(b)), "progress")
        return int(a.value), int(b.value)

    def last_propagate_ms(self) -> float:
        v = C.c_double()
        self._check(self._L.tardis_mc_last_propagate_ms(self._h, C.byref(v)), "last_propagate_ms")
        return v.value

    def last_kernel_times(self) -> dict:
        """Device time (ms) of the seeding and propagation launches of the last propagate(), and the launch count."""
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        self._check(self._L.tardis_mc_last_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(n)), "last_kernel_times")
        e = C.c_double()
        self._check(self._L.tardis_mc_last_estimator_ms(self._h, C.byref(e)), "last_estimator_ms")
        return {"seed_ms": a.value, "propagate_ms": b.value, "launches": n.value, "estimator_ms": e.value}

    def last_compactions(self) -> int:
        """How often the last propagate() packed the live lanes of its drain into fewer waves (option drain_compact)."""
        return int(self._L.tardis_mc_last_compactions(self._h))

    def last_variant(self) -> int:
        """Propagation kernel of the last propagate(): 0 lane, 1 group, 2 wave + group sweeps, 3 wave + lane sweeps,
        4 wave + volley queue (v-packets traced by vpacket_trace_kernel between its launches)."""
        return int(self._L.tardis_mc_last_variant(self._h))

    def get_results(self, output_nus=None, output_energies=None, track_last_interaction=True,
                    want_line_estimators=True, vpacket_log_capacity=None, want_packet_outputs=True, trackers=None) -> _abi.ResultBuffers:
        """Copy results out.  Every part is optional: per-packet outputs (`want_packet_outputs`), the last-interaction
        trackers, the [L,S] line estimators -- whatever is not asked for stays on the device (the resident outer-iteration
        path reads the spectrum and the radiation field through packet_spectrum() / radiation_field() instead)."""
        # (`trackers`: the caller's own LastInteractionTrackers -- the library writes straight into its arrays instead of into fresh ones that
        # the caller would then copy: 1.1 GB allocated, touched and copied once more per 1e7 packets otherwise)
        if not track_last_interaction:
            trackers = None
        elif not (isinstance(trackers, st.LastInteractionTrackers) and len(trackers) == self.n_packets):
            trackers = st.LastInteractionTrackers(self.n_packets)
        cap = 0
        if self._vpk_log:
            cap = int(vpacket_log_capacity if vpacket_log_capacity is not None else self.n_packets * self._n_v * 64)
        res = _abi.ResultBuffers(self.n_packets, self.n_shells, self.n_lines, self.n_grid, output_nus, output_energies,
                                 trackers, cap, want_line_estimators, want_packet_outputs)
        rc = self._L.tardis_mc_get_results(self._h, res.ref())
        self._check(rc, "get_results", int(res.struct.first_error_packet))
        return res

    def stream_results(self, output_nus=None, output_energies=None, trackers=None) -> None:
        """Register the host arrays the NEXT propagate call may fill while it runs (`tardis_mc_stream_results`): a long call is several
        launches, and the per-packet results of the packets a launch has finished are copied beside the following launch instead of
        after the last one.  get_results() on the same arrays then sends only what is missing; on other arrays it copies everything.
        No arguments: disarm."""
        if output_nus is None and output_energies is None and trackers is None:
            self._stream_keep = None
            self._check(self._L.tardis_mc_stream_results(self._h, None), "stream_results")
            return
        if isinstance(trackers, st.LastInteractionTrackers) and len(trackers) != self.n_packets:
            raise ValueError("trackers must hold n_packets entries")
        res = _abi.ResultBuffers(self.n_packets, 0, 0, 0, output_nus, output_energies, trackers, 0, False, output_nus is not None)
        self._stream_keep = res  # (the arrays stay alive until the call that fills them is over)
        self._check(self._L.tardis_mc_stream_results(self._h, res.ref()), "stream_results")

    def streamed_packets(self) -> tuple[int, int]:
        """(packets whose results the last propagate call copied out while it ran, how many of them get_results sends again)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._L.tardis_mc_streamed_packets(self._h, C.byref(a), C.byref(b)), "streamed_packets")
        return int(a.value), int(b.value)

    def run(self, packet_collection, geometry, time_explosion, opacity_state, montecarlo_configuration, spectrum_frequency_grid,
            number_of_vpackets=None, track_last_interaction=True, vpacket_log_capacity=None) -> _abi.ResultBuffers:
        """The one-shot form of the boundary, `tardis_mc_run` (include/tardis_mc.h): geometry, opacity, configuration and packets in,
        one propagation, results out -- the call a ctypes binding inside `run_classic` makes when nothing is to stay resident
        (modes/classic/solver.py:223-234).  `vpacket_log_capacity`: entries of the caller's v-packet log arrays for THIS call (None:
        sized like get_results does; the library restores the context's own setting when the call returns, however it ends)."""
        mp = _abi.marshal_packets(packet_collection)
        mg = _abi.marshal_geometry(geometry, time_explosion)
        mo = _abi.marshal_opacity(opacity_state)
        mc = _abi.marshal_config(montecarlo_configuration, spectrum_frequency_grid, number_of_vpackets)
        P, S, L = int(mp.struct.n_packets), int(mg.struct.n_shells), int(mo.struct.n_lines)
        n_v = int(mc.struct.number_of_vpackets)
        log = bool(mc.struct.enable_vpacket_tracking) and n_v > 0
        cap = (int(vpacket_log_capacity) if vpacket_log_capacity is not None else P * n_v * 64) if log else 0
        trackers = st.LastInteractionTrackers(P) if track_last_interaction else None
        res = _abi.ResultBuffers(P, S, L, int(mc.struct.n_spectrum_grid), None, None, trackers, cap)
        self.resident_opacity = None
        self.results_generation += 1
        self.estimators_generation += 1
        self.packets_generation += 1
        rc = self._L.tardis_mc_run(self._h, mp.ref(), mg.ref(), mo.ref(), mc.ref(), res.ref())
        self.n_packets, self.n_shells, self.n_lines, self.n_grid = P, S, L, int(mc.struct.n_spectrum_grid)
        self._n_v, self._vpk_log = n_v, log
        self._check(rc, "run", int(res.struct.first_error_packet))
        self.resident_opacity = opacity_state
        self.results_generation += 1
        self.estimators_generation += 1
        return res

    def create_blackbody_packets(self, n_packets: int, radius: float, temperature: float, base_seed: int = 23111963,
                                 seed_offset: int = 0, *, first: int = 0, count: int | None = None,
                                 max_seed_val: int = 2**32 - 1, l_samples: int = 1000) -> None:
        """BlackBodySimpleSource.create_packets on the device (packet_source/base.py:195-253): the packets
        np.random.default_rng(base_seed + seed_offset) would have produced, slice [first, first+count), left resident
        as this engine's packet inputs."""
        count = n_packets - first if count is None else count
        st = (C.c_uint64 * 4)()
        self._check(self._L.tardis_mc_pcg64_seed(int(base_seed + seed_offset), st), "pcg64_seed")
        l_array = np.cumsum(np.arange(1, l_samples, dtype=np.float64) ** -4)  # black_body.py:174
        self._check(self._L.tardis_mc_create_blackbody_packets(self._h, int(n_packets), int(first), int(count), float(radius),
                                                               float(temperature), st, int(max_seed_val),
                                                               l_array.ctypes.data, len(l_array)), "create_blackbody_packets")
        self.n_packets = int(count)
        self.packets_generation += 1

    def get_packets(self) -> dict:
        n = self.n_packets
        out = {k: np.empty(n) for k in ("initial_radii", "initial_nus", "initial_mus", "initial_energies")}
        out["packet_seeds"] = np.empty(n, dtype=np.int64)
        self._check(self._L.tardis_mc_get_packets(self._h, *(out[k].ctypes.data for k in
                                                             ("initial_radii", "initial_nus", "initial_mus", "initial_energies",
                                                              "packet_seeds"))), "get_packets")
        return out

    def packet_spectrum(self, time_of_simulation: float, luminosity_nu_start: float = 0.0,
                        luminosity_nu_end: float = float("inf")) -> dict:
        """Real-packet spectrum and filtered luminosities computed on the device from the resident packet outputs
        (what SpectrumSolver.montecarlo_emitted/reabsorbed_luminosity and calculate_filtered_luminosity return)."""
        B = self.n_grid - 1
        he, hr = np.zeros(B), np.zeros(B)
        le, lr = C.c_double(), C.c_double()
        self._check(self._L.tardis_mc_packet_spectrum(self._h, float(time_of_simulation), float(luminosity_nu_start),
                                                      float(luminosity_nu_end), he.ctypes.data, hr.ctypes.data,
                                                      C.byref(le), C.byref(lr)), "packet_spectrum")
        return {"montecarlo_emitted_luminosity": he, "montecarlo_reabsorbed_luminosity": hr,
                "emitted_luminosity": le.value, "reabsorbed_luminosity": lr.value}

    def radiation_field(self, time_of_simulation: float, volume, w_epsilon: float = 1e-10,
                        detailed_optical_window: bool = False, want_j_blues: bool = True) -> dict:
        """MCRadiationFieldPropertiesSolver.solve (estimators/mc_rad_field_solver.py:37-144) on the resident estimators."""
        volume = np.ascontiguousarray(volume, dtype=np.float64)
        if volume.shape != (self.n_shells,):
            raise ValueError("volume must have one entry per shell")
        t_rad, w = np.empty(self.n_shells), np.empty(self.n_shells)
        jb = np.empty((self.n_lines, self.n_shells)) if want_j_blues else None
        self._check(self._L.tardis_mc_radiation_field(self._h, float(time_of_simulation), volume.ctypes.data, float(w_epsilon),
                                                      int(detailed_optical_window), t_rad.ctypes.data, w.ctypes.data,
                                                      jb.ctypes.data if jb is not None else None), "radiation_field")
        return {"t_radiative": t_rad, "dilution_factor": w, "j_blues": jb}

    def last_counters(self) -> dict:
        out = (C.c_int64 * len(_abi.COUNTER_NAMES))()
        self._check(self._L.tardis_mc_last_counters(self._h, out), "last_counters")
        return dict(zip(_abi.COUNTER_NAMES, [int(v) for v in out]))

    def formal_integral(self, inner_temperature: float, frequencies, att_S_ul, Jred_lu, Jblue_lu, n_impact_parameters: int = 1000,
                        want_intensities: bool = False):
        """numba_formal_integral (spectrum/formal_integral/formal_integral_numba.py:375-560) on the resident geometry / opacity:
        returns (luminosity_densities, intensities_nu_p or None)."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64).ravel()
        freqs, att, jred, jblue = f(frequencies), f(att_S_ul), f(Jred_lu), f(Jblue_lu)
        n = self.n_shells * self.n_lines
        if att.size != n or jred.size != n or jblue.size != n:
            raise ValueError("att_S_ul / Jred_lu / Jblue_lu must have n_shells * n_lines entries (shell-major)")
        lum = np.empty(freqs.size)
        inten = np.empty((freqs.size, int(n_impact_parameters))) if want_intensities else None
        self._check(self._L.tardis_mc_formal_integral(self._h, float(inner_temperature), freqs.ctypes.data, freqs.size, att.ctypes.data,
                                                      jred.ctypes.data, jblue.ctypes.data, int(n_impact_parameters), lum.ctypes.data,
                                                      inten.ctypes.data if inten is not None else None), "formal_integral")
        return lum, inten

    # -- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * _abi.UNIQUE_ID_BYTES)()
        rc = _lib.lib().tardis_mc_comm_get_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"tardis_mc_comm_get_unique_id failed ({rc})")
        return bytes(buf)

    def comm_init(self, rank: int, world_size: int, unique_id: bytes):
        buf = (C.c_uint8 * _abi.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._L.tardis_mc_comm_init(self._h, rank, world_size, buf), "comm_init")

    def comm_check(self) -> int:
        """Self-check of the RCCL communicator: the one-element all-reduce of (rank + 1) came back as N (N + 1) / 2; returns N."""
        n = C.c_int(0)
        self._check(self._L.tardis_mc_comm_check(self._h, C.byref(n)), "comm_check")
        return int(n.value)

    def allreduce_estimators(self):
        self.estimators_generation += 1
        self._check(self._L.tardis_mc_allreduce_estimators(self._h), "allreduce_estimators")

    # -- diagnostics
    def debug_eval(self, op: int, x, y=None, n=None) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = int(x.size if n is None else n)
        out = np.empty(n)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            yp = y.ctypes.data
        self._check(self._L.tardis_mc_debug_eval(self._h, op, x.ctypes.data, yp, out.ctypes.data, n), "debug_eval")
        return out


    def debug_microbench(self, which: int, n_doubles: int, iters: int, blocks: int) -> float:
        v = C.c_double()
        self._check(self._L.tardis_mc_debug_microbench(self._h, which, n_doubles, iters, blocks, C.byref(v)), "microbench")
        return v.value


def device_count() -> int:
    return int(_lib.lib().tardis_mc_device_count())
