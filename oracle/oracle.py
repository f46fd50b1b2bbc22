"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/libtardis_mc_oracle.so).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package
(tardis_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import types

import numpy as np

from tardis_amd import _abi
from tardis_amd import state as st

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtardis_mc_oracle.so")
_lib = None

MATH_LIBM, MATH_PORTABLE = 0, 1


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc, IEEE-strict flags)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_mc_run.restype = C.c_int
        L.oracle_mc_run.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int]
        L.oracle_log.restype = C.c_double
        L.oracle_log.argtypes = [C.c_double, C.c_int]
        L.oracle_exp.restype = C.c_double
        L.oracle_exp.argtypes = [C.c_double, C.c_int]
        for f in (L.oracle_log_array, L.oracle_exp_array):
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.oracle_mt19937_random.restype = None
        L.oracle_mt19937_random.argtypes = [C.c_uint32, C.c_int64, C.c_void_p]
        for f in (L.oracle_doppler_factor, L.oracle_inverse_doppler_factor):
            f.restype = C.c_double
            f.argtypes = [C.c_double, C.c_double, C.c_int]
        for f in (L.oracle_angle_aberration_cmf_to_lf, L.oracle_angle_aberration_lf_to_cmf):
            f.restype = C.c_double
            f.argtypes = [C.c_double] * 3
        L.oracle_distance_boundary.restype = None
        L.oracle_distance_boundary.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_void_p]
        L.oracle_distance_line.restype = C.c_int
        L.oracle_distance_line.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.oracle_packet_step.restype = C.c_int
        L.oracle_packet_step.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double] + [C.c_void_p] * 4 + [
            C.c_int, C.c_void_p]
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def max_threads(cap: int = 64) -> int:
    """Threads worth using for the OpenMP oracle: OpenMP's maximum, clipped by the CPU affinity mask, the cgroup CPU
    quota of the container (oversubscribing a quota-limited pod is far slower than using fewer threads) and `cap`."""
    n = int(lib().oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def run(packet_collection, geometry, time_explosion, opacity_state, montecarlo_configuration,
        spectrum_frequency_grid, number_of_vpackets=None, math_mode=MATH_LIBM, n_threads=1,
        track_last_interaction=True, write_outputs_in_place=False, sigma_thomson=None):
    """Run the oracle main loop; returns a namespace with the reference's result names."""
    pk = _abi.marshal_packets(packet_collection)
    geo = _abi.marshal_geometry(geometry, time_explosion)
    op = _abi.marshal_opacity(opacity_state)
    cfg = _abi.marshal_config(montecarlo_configuration, spectrum_frequency_grid, number_of_vpackets, sigma_thomson)
    P = pk.struct.n_packets
    trackers = st.LastInteractionTrackers(P) if track_last_interaction else None
    cap = 0
    if cfg.struct.enable_vpacket_tracking and cfg.struct.number_of_vpackets > 0:
        cap = int(P * cfg.struct.number_of_vpackets * 64)
    out_nu = packet_collection.output_nus if write_outputs_in_place else None
    out_e = packet_collection.output_energies if write_outputs_in_place else None
    res = _abi.ResultBuffers(P, op.struct.n_shells, op.struct.n_lines, cfg.struct.n_spectrum_grid, out_nu, out_e,
                             trackers, cap)
    rc = lib().oracle_mc_run(pk.ref(), geo.ref(), op.ref(), cfg.ref(), res.ref(), int(math_mode), int(n_threads))
    n_log = min(res.vpacket_log_count, cap)
    return types.SimpleNamespace(
        return_code=rc, error_code=int(res.struct.error_code), first_error_packet=int(res.struct.first_error_packet),
        output_nus=res.output_nus, output_energies=res.output_energies,
        j_estimator=res.j_estimator, nu_bar_estimator=res.nu_bar_estimator,
        j_blue_estimator=res.j_blue_estimator, edotlu_estimator=res.edotlu_estimator,
        v_packets_energy_hist=res.v_packets_energy_hist, trackers=trackers, counters=res.counters,
        vpacket_nus=res.vpacket_nus[:n_log], vpacket_energies=res.vpacket_energies[:n_log],
        vpacket_initial_mus=res.vpacket_initial_mus[:n_log], vpacket_initial_rs=res.vpacket_initial_rs[:n_log],
        vpacket_log_count=res.vpacket_log_count)


def mt19937_random(seed: int, n: int) -> np.ndarray:
    out = np.empty(n)
    lib().oracle_mt19937_random(seed, n, out.ctypes.data)
    return out


def log_array(x, math_mode):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().oracle_log_array(x.ctypes.data, y.ctypes.data, x.size, math_mode)
    return y


def exp_array(x, math_mode):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().oracle_exp_array(x.ctypes.data, y.ctypes.data, x.size, math_mode)
    return y


def distance_boundary(r, mu, r_inner, r_outer):
    d = C.c_double()
    delta = C.c_int64()
    lib().oracle_distance_boundary(r, mu, r_inner, r_outer, C.byref(d), C.byref(delta))
    return d.value, delta.value


def distance_line(nu, r, mu, comov_nu, is_last_line, nu_line, time_explosion, full_relativity=False):
    d = C.c_double()
    rc = lib().oracle_distance_line(nu, r, mu, comov_nu, int(is_last_line), nu_line, time_explosion,
                                    int(full_relativity), C.byref(d))
    return rc, d.value


STEP_TRACE_PACKET, STEP_MOVE, STEP_THOMSON, STEP_LINE_SCATTER, STEP_CROSS_SHELL, STEP_VOLLEY = range(6)


def packet_step(what, packet, ids, seed, arg, geometry, opacity_state, montecarlo_configuration,
                spectrum_frequency_grid=None, math_mode=MATH_LIBM, time_explosion=None, sigma_thomson=None):
    """Single leaf call on one packet.  packet = [r, mu, nu, energy]; ids = [next_line_id, shell, status]."""
    grid = np.zeros(2) if spectrum_frequency_grid is None else spectrum_frequency_grid
    geo = _abi.marshal_geometry(geometry, time_explosion)
    op = _abi.marshal_opacity(opacity_state)
    cfg = _abi.marshal_config(montecarlo_configuration, grid, None, sigma_thomson)
    cap = max(int(cfg.struct.number_of_vpackets), 1)
    res = _abi.ResultBuffers(0, op.struct.n_shells, op.struct.n_lines, len(grid), vpacket_log_capacity=cap)
    pkt = np.array(packet, dtype=np.float64)
    idv = np.array(ids, dtype=np.int64)
    out = np.zeros(3)
    rc = lib().oracle_packet_step(what, pkt.ctypes.data, idv.ctypes.data, seed, float(arg), geo.ref(), op.ref(),
                                  cfg.ref(), res.ref(), math_mode, out.ctypes.data)
    n = min(res.vpacket_log_count, cap)
    return types.SimpleNamespace(return_code=rc, packet=pkt, ids=idv, distance=out[0], interaction_type=int(out[1]),
                                 delta_shell=int(out[2]), j_estimator=res.j_estimator,
                                 nu_bar_estimator=res.nu_bar_estimator, j_blue_estimator=res.j_blue_estimator,
                                 edotlu_estimator=res.edotlu_estimator, vpacket_nus=res.vpacket_nus[:n],
                                 vpacket_energies=res.vpacket_energies[:n], counters=res.counters)
