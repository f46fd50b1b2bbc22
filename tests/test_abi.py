"""CPU-side checks of the C ABI library: it loads without a GPU, exports every symbol the header declares,
and fails loudly (no fallback) when no device is present."""
import os
import re

import numpy as np
import pytest

from tardis_amd import _abi, _lib, state as st, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "tardis_mc.h")).read()
    declared = set(re.findall(r"\b(tardis_mc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_struct_sizes(lib):
    assert lib.tardis_mc_abi_version() == _abi.ABI_VERSION
    import ctypes as C
    assert C.sizeof(_abi.TardisMcConfig) == 4 * 4 + 8 + 5 * 8 + 8 + 8
    assert C.sizeof(_abi.TardisMcPackets) == 6 * 8
    assert C.sizeof(_abi.TardisMcGeometry) == 4 * 8
    assert C.sizeof(_abi.TardisMcOpacity) == 13 * 8
    assert C.sizeof(_abi.TardisMcResult) == (7 + 9 + 5 + 2 + 4) * 8 + 8 * 8 + 8 + 4 + 4


def test_no_gpu_means_loud_failure(lib):
    if lib.tardis_mc_device_count() > 0:
        pytest.skip("a GPU is visible")
    from tardis_amd.engine import Engine
    with pytest.raises(_lib.EngineUnavailable):
        Engine(0)
    from tardis_amd import transport
    prob = synthetic.make_problem(n_packets=8, n_lines=50, n_shells=3)
    with pytest.raises(_lib.EngineUnavailable):
        transport.montecarlo_transport_with_vpackets(
            prob.packet_collection, prob.geometry, prob.time_explosion, prob.opacity_state,
            prob.montecarlo_configuration, prob.spectrum_frequency_grid, None, 0, False, None)


def test_marshalling_handles_strided_shell_slices():
    prob = synthetic.make_problem(n_packets=4, n_lines=40, n_shells=6, line_interaction_type="macroatom")
    sliced = prob.opacity_state[1:4]            # the reference slices shells the same way (solver.py:127-129)
    assert not sliced.tau_sobolev.flags.c_contiguous
    m = _abi.marshal_opacity(sliced)
    assert (m.struct.n_lines, m.struct.n_shells, m.struct.n_transitions) == (40, 3, 120)
    assert np.array_equal(np.ctypeslib.as_array(m.struct.tau_sobolev, (40, 3)), prob.opacity_state.tau_sobolev[:, 1:4])


def test_packet_collection_shard_views():
    prob = synthetic.make_problem(n_packets=10, n_lines=20, n_shells=2)
    pc = prob.packet_collection
    parts = [pc.shard(r, 3) for r in range(3)]
    assert sum(p.number_of_packets for p in parts) == 10
    parts[1].output_nus[:] = 7.0
    lo, hi = (1 * 10) // 3, (2 * 10) // 3
    assert np.all(pc.output_nus[lo:hi] == 7.0) and np.all(pc.output_nus[:lo] == -99.0)
