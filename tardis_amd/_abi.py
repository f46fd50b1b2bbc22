"""ctypes mirror of include/tardis_mc.h (struct layouts + marshalling from the Python state objects).

Only interface definitions live here: no library is loaded by importing this module.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import state as st

ABI_VERSION = 2
UNIQUE_ID_BYTES = 128
N_COUNTERS = 8
COUNTER_NAMES = ("line_visits", "events", "macro_transitions", "vpacket_line_visits", "vpackets", "rng_draws",
                 "packets", "reserved")

ERR_INVALID_ARGUMENT, ERR_HIP, ERR_MONTECARLO, ERR_MACRO_ATOM, ERR_UNSUPPORTED, ERR_COMM, ERR_STATE = (
    -1, -2, -3, -4, -5, -6, -7)

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int64)


class TardisMcConfig(C.Structure):
    _fields_ = [
        ("enable_full_relativity", C.c_int32),
        ("line_interaction_type", C.c_int32),
        ("disable_line_scattering", C.c_int32),
        ("enable_vpacket_tracking", C.c_int32),
        ("number_of_vpackets", C.c_int64),
        ("survival_probability", C.c_double),
        ("vpacket_tau_russian", C.c_double),
        ("vpacket_spawn_start_frequency", C.c_double),
        ("vpacket_spawn_end_frequency", C.c_double),
        ("sigma_thomson", C.c_double),
        ("n_spectrum_grid", C.c_int64),
        ("spectrum_frequency_grid", _pd),
    ]


class TardisMcPackets(C.Structure):
    _fields_ = [
        ("n_packets", C.c_int64),
        ("initial_radii", _pd),
        ("initial_nus", _pd),
        ("initial_mus", _pd),
        ("initial_energies", _pd),
        ("packet_seeds", _pi),
    ]


class TardisMcGeometry(C.Structure):
    _fields_ = [
        ("n_shells", C.c_int64),
        ("r_inner", _pd),
        ("r_outer", _pd),
        ("time_explosion", C.c_double),
    ]


class TardisMcOpacity(C.Structure):
    _fields_ = [
        ("n_lines", C.c_int64),
        ("n_shells", C.c_int64),
        ("n_transitions", C.c_int64),
        ("n_macro_block_edges", C.c_int64),
        ("electron_density", _pd),
        ("line_list_nu", _pd),
        ("tau_sobolev", _pd),
        ("transition_probabilities", _pd),
        ("line2macro_level_upper", _pi),
        ("macro_block_edge_index", _pi),
        ("transition_type", _pi),
        ("destination_level_id", _pi),
        ("transition_line_id", _pi),
    ]


_LI_F64 = ("li_radius", "li_nu", "li_energy", "li_before_nu", "li_before_mu", "li_before_energy", "li_after_nu",
           "li_after_mu", "li_after_energy")
_LI_I64 = ("li_shell_id", "li_interaction_type", "li_line_absorb_id", "li_line_emit_id", "li_interactions_count")


class TardisMcResult(C.Structure):
    _fields_ = (
        [(n, _pd) for n in ("output_nus", "output_energies", "j_estimator", "nu_bar_estimator", "j_blue_estimator",
                            "edotlu_estimator", "v_packets_energy_hist")]
        + [(n, _pd) for n in _LI_F64]
        + [(n, _pi) for n in _LI_I64]
        + [("vpacket_log_capacity", C.c_int64), ("vpacket_log_count", C.c_int64)]
        + [(n, _pd) for n in ("vpacket_nus", "vpacket_energies", "vpacket_initial_mus", "vpacket_initial_rs")]
        + [("counters", C.c_int64 * N_COUNTERS), ("first_error_packet", C.c_int64), ("error_code", C.c_int32),
           ("reserved", C.c_int32)]
    )


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_pd)


def _ip(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(_pi)


class Marshalled:
    """A ctypes struct plus the numpy arrays that back its pointers (kept alive together)."""

    def __init__(self, struct, keep):
        self.struct = struct
        self.keep = keep

    def ref(self):
        return C.byref(self.struct)


def marshal_packets(pc) -> Marshalled:
    arrs = [np.ascontiguousarray(getattr(pc, n), dtype=np.float64)
            for n in ("initial_radii", "initial_nus", "initial_mus", "initial_energies")]
    seeds = np.ascontiguousarray(pc.packet_seeds, dtype=np.int64)
    n = len(arrs[0])
    if not all(len(a) == n for a in arrs) or len(seeds) != n:
        raise ValueError("packet arrays must have equal length")
    s = TardisMcPackets(n, _dp(arrs[0]), _dp(arrs[1]), _dp(arrs[2]), _dp(arrs[3]), _ip(seeds))
    return Marshalled(s, arrs + [seeds])


def marshal_geometry(geometry, time_explosion=None) -> Marshalled:
    r_in = np.ascontiguousarray(geometry.r_inner, dtype=np.float64)
    r_out = np.ascontiguousarray(geometry.r_outer, dtype=np.float64)
    t = float(geometry.time_explosion if time_explosion is None else time_explosion)
    if len(r_in) != len(r_out):
        raise ValueError("r_inner / r_outer length mismatch")
    return Marshalled(TardisMcGeometry(len(r_in), _dp(r_in), _dp(r_out), t), [r_in, r_out])


def marshal_opacity(op) -> Marshalled:
    ne = np.ascontiguousarray(op.electron_density, dtype=np.float64)
    nu = np.ascontiguousarray(op.line_list_nu, dtype=np.float64)
    tau = np.ascontiguousarray(op.tau_sobolev, dtype=np.float64)  # shell slices are strided views
    prob = np.ascontiguousarray(op.transition_probabilities, dtype=np.float64)
    l2m = np.ascontiguousarray(op.line2macro_level_upper, dtype=np.int64)
    edge = np.ascontiguousarray(op.macro_block_edge_index, dtype=np.int64)
    ttype = np.ascontiguousarray(op.transition_type, dtype=np.int64)
    dest = np.ascontiguousarray(op.destination_level_id, dtype=np.int64)
    tline = np.ascontiguousarray(op.transition_line_id, dtype=np.int64)
    L, S = tau.shape
    if len(nu) != L or len(ne) != S:
        raise ValueError("tau_sobolev must be [n_lines, n_shells]")
    if prob.ndim == 2 and prob.shape == (1, 1) and S > 1 and len(edge) <= 1 and len(ttype) <= 1:
        # line_interaction_type "scatter": OpacityState.to_numba passes np.zeros((1, 1)) and size-1 macro tables
        # (tardis/opacities/opacity_state.py:199-209); the engine wants one column per shell
        prob = np.zeros((1, S))
    T = prob.shape[0]
    if prob.ndim != 2 or prob.shape[1] != S:
        raise ValueError("transition_probabilities must be [n_transitions, n_shells]")
    s = TardisMcOpacity(L, S, T, len(edge), _dp(ne), _dp(nu), _dp(tau), _dp(prob), _ip(l2m), _ip(edge), _ip(ttype),
                        _ip(dest), _ip(tline))
    return Marshalled(s, [ne, nu, tau, prob, l2m, edge, ttype, dest, tline])


def marshal_config(cfg, spectrum_frequency_grid, number_of_vpackets=None, sigma_thomson=None) -> Marshalled:
    grid = np.ascontiguousarray(spectrum_frequency_grid, dtype=np.float64)
    if sigma_thomson is None:
        # modes/classic/solver.py:291-300: disable_electron_scattering -> sigma_T = 1e-200
        sigma_thomson = 1e-200 if getattr(cfg, "DISABLE_ELECTRON_SCATTERING", False) else st.SIGMA_THOMSON
    n_v = int(cfg.NUMBER_OF_VPACKETS if number_of_vpackets is None else number_of_vpackets)
    s = TardisMcConfig(
        int(bool(cfg.ENABLE_FULL_RELATIVITY)), int(cfg.LINE_INTERACTION_TYPE), int(bool(cfg.DISABLE_LINE_SCATTERING)),
        int(bool(cfg.ENABLE_VPACKET_TRACKING)), n_v, float(cfg.SURVIVAL_PROBABILITY), float(cfg.VPACKET_TAU_RUSSIAN),
        float(cfg.VPACKET_SPAWN_START_FREQUENCY), float(cfg.VPACKET_SPAWN_END_FREQUENCY), float(sigma_thomson),
        len(grid), _dp(grid))
    return Marshalled(s, [grid])


class ResultBuffers:
    """Owns (or borrows) the output arrays and exposes them through a TardisMcResult struct."""

    def __init__(self, n_packets, n_shells, n_lines, n_grid, output_nus=None, output_energies=None,
                 trackers: st.LastInteractionTrackers | None = None, vpacket_log_capacity=0,
                 want_line_estimators=True, want_packet_outputs=True):
        P, S, L = int(n_packets), int(n_shells), int(n_lines)
        if want_packet_outputs:
            self.output_nus = output_nus if output_nus is not None else np.full(P, -99.0)
            self.output_energies = output_energies if output_energies is not None else np.full(P, -99.0)
            for a in (self.output_nus, self.output_energies):
                if a.dtype != np.float64 or not a.flags.c_contiguous or len(a) != P:
                    raise ValueError("output arrays must be contiguous float64 of length n_packets")
        else:  # the per-packet results stay on the device (NULL pointers: the library skips those copies)
            self.output_nus = self.output_energies = None
        self.j_estimator = np.zeros(S)
        self.nu_bar_estimator = np.zeros(S)
        self.j_blue_estimator = np.zeros((L, S)) if want_line_estimators else None
        self.edotlu_estimator = np.zeros((L, S)) if want_line_estimators else None
        self.v_packets_energy_hist = np.zeros(int(n_grid))
        self.trackers = trackers
        cap = int(vpacket_log_capacity)
        self.vpacket_nus = np.empty(cap)
        self.vpacket_energies = np.empty(cap)
        self.vpacket_initial_mus = np.empty(cap)
        self.vpacket_initial_rs = np.empty(cap)
        r = TardisMcResult()
        if want_packet_outputs:
            r.output_nus = _dp(self.output_nus)
            r.output_energies = _dp(self.output_energies)
        r.j_estimator = _dp(self.j_estimator)
        r.nu_bar_estimator = _dp(self.nu_bar_estimator)
        if want_line_estimators:
            r.j_blue_estimator = _dp(self.j_blue_estimator)
            r.edotlu_estimator = _dp(self.edotlu_estimator)
        r.v_packets_energy_hist = _dp(self.v_packets_energy_hist)
        if trackers is not None:
            m = {"li_radius": "radius", "li_nu": "nu", "li_energy": "energy", "li_before_nu": "before_nu",
                 "li_before_mu": "before_mu", "li_before_energy": "before_energy", "li_after_nu": "after_nu",
                 "li_after_mu": "after_mu", "li_after_energy": "after_energy"}
            for k, v in m.items():
                setattr(r, k, _dp(getattr(trackers, v)))
            m = {"li_shell_id": "shell_id", "li_interaction_type": "interaction_type",
                 "li_line_absorb_id": "interaction_line_absorb_id", "li_line_emit_id": "interaction_line_emit_id",
                 "li_interactions_count": "interactions_count"}
            for k, v in m.items():
                setattr(r, k, _ip(getattr(trackers, v)))
        r.vpacket_log_capacity = cap
        if cap:
            r.vpacket_nus = _dp(self.vpacket_nus)
            r.vpacket_energies = _dp(self.vpacket_energies)
            r.vpacket_initial_mus = _dp(self.vpacket_initial_mus)
            r.vpacket_initial_rs = _dp(self.vpacket_initial_rs)
        r.first_error_packet = -1
        self.struct = r

    def ref(self):
        return C.byref(self.struct)

    @property
    def counters(self) -> dict:
        return dict(zip(COUNTER_NAMES, [int(v) for v in self.struct.counters]))

    @property
    def vpacket_log_count(self) -> int:
        return int(self.struct.vpacket_log_count)
