"""tardis_amd -- MI355X-native Monte Carlo packet-propagation engine for TARDIS (drop-in for one path).

Scope: the reference's ``montecarlo_transport_with_vpackets`` / ``packet_propagation`` path,
re-designed as hand-written HIP kernels for gfx950 behind a C ABI (include/tardis_mc.h), reached from Python
through ctypes.  Nothing else of TARDIS is rebuilt here.
"""
from . import state  # noqa: F401

__version__ = "0.1.0"
