"""Loader of libtardis_mc_hip.so (the HIP engine behind include/tardis_mc.h).

The library is built in-tree (tardis_amd/csrc/Makefile -> tardis_amd/libtardis_mc_hip.so).  There is NO CPU
fallback: if the library is missing or no MI355X is visible, the transport entry points raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TARDIS_MC_LIB: a profiling build of the same sources, e.g. with -DTMC_SECTION_TIMERS; tools/ only)
LIB_PATH = os.environ.get("TARDIS_MC_LIB") or os.path.join(_HERE, "libtardis_mc_hip.so")
_lib = None

# every symbol include/tardis_mc.h declares: name -> (restype, argtypes)
_vp, _i, _ll = C.c_void_p, C.c_int, C.c_longlong
SYMBOLS = {
    "tardis_mc_abi_version": (_i, []),
    "tardis_mc_device_count": (_i, []),
    "tardis_mc_create": (_i, [_i, C.POINTER(_vp)]),
    "tardis_mc_destroy": (None, [_vp]),
    "tardis_mc_last_error": (C.c_char_p, [_vp]),
    "tardis_mc_set_option": (_i, [_vp, C.c_char_p, _ll]),
    "tardis_mc_set_geometry": (_i, [_vp, _vp]),
    "tardis_mc_set_opacity": (_i, [_vp, _vp]),
    "tardis_mc_set_config": (_i, [_vp, _vp]),
    "tardis_mc_set_packets": (_i, [_vp, _vp]),
    "tardis_mc_reset_estimators": (_i, [_vp]),
    "tardis_mc_propagate": (_i, [_vp]),
    "tardis_mc_synchronize": (_i, [_vp]),
    "tardis_mc_last_propagate_ms": (_i, [_vp, C.POINTER(C.c_double)]),
    "tardis_mc_last_kernel_times": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "tardis_mc_last_estimator_ms": (_i, [_vp, C.POINTER(C.c_double)]),
    "tardis_mc_last_counters": (_i, [_vp, C.POINTER(C.c_int64)]),
    "tardis_mc_last_variant": (_i, [_vp]),
    "tardis_mc_last_compactions": (_i, [_vp]),
    "tardis_mc_progress": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tardis_mc_get_results": (_i, [_vp, _vp]),
    "tardis_mc_pcg64_seed": (_i, [C.c_uint64, C.POINTER(C.c_uint64)]),
    "tardis_mc_create_blackbody_packets": (_i, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                                C.POINTER(C.c_uint64), C.c_uint32, _vp, C.c_int64]),
    "tardis_mc_get_packets": (_i, [_vp] * 6),
    "tardis_mc_packet_spectrum": (_i, [_vp, C.c_double, C.c_double, C.c_double, _vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tardis_mc_radiation_field": (_i, [_vp, C.c_double, _vp, C.c_double, C.c_int, _vp, _vp, _vp]),
    "tardis_mc_formal_integral": (_i, [_vp, C.c_double, _vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "tardis_mc_stream_results": (_i, [_vp, _vp]),
    "tardis_mc_streamed_packets": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tardis_mc_run": (_i, [_vp] * 6),
    "tardis_mc_comm_get_unique_id": (_i, [_vp]),
    "tardis_mc_comm_init": (_i, [_vp, _i, _i, _vp]),
    "tardis_mc_allreduce_estimators": (_i, [_vp]),
    "tardis_mc_comm_check": (_i, [_vp, C.POINTER(_i)]),
    "tardis_mc_debug_eval": (_i, [_vp, _i, _vp, _vp, _vp, C.c_int64]),
    "tardis_mc_debug_microbench": (_i, [_vp, _i, C.c_int64, _i, _i, C.POINTER(C.c_double)]),
}


class EngineUnavailable(RuntimeError):
    """The HIP engine cannot be used (library not built, or no GPU)."""


def build(force: bool = False) -> str:
    """Compile the HIP engine for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libtardis_mc_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                f"{LIB_PATH} is missing: build it with `make -C tardis_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback for the transport engine.")
        L = C.CDLL(LIB_PATH)
        L.tardis_mc_abi_version.restype = _i
        other_build = bool(os.environ.get("TARDIS_MC_LIB"))  # (tools/ A/B against an older build: what it lacks stays unbound)
        if L.tardis_mc_abi_version() != _abi.ABI_VERSION and not other_build:  # (before the symbols are bound: an older build lacks the newer ones)
            raise EngineUnavailable(f"{LIB_PATH}: ABI version {L.tardis_mc_abi_version()}, this package needs {_abi.ABI_VERSION}; rebuild it")
        for name, (res, args) in SYMBOLS.items():
            if other_build and not hasattr(L, name):
                continue
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
