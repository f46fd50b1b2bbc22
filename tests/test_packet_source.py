"""Packet source: the oracle's restatement of NumPy's PCG64 streams against NumPy itself, the C-ABI
seed expansion against NumPy (host code, no GPU), and -- on the GPU -- the device source against a host run."""
import ctypes as C

import numpy as np
import pytest

from oracle import pcg64_source as src
from tardis_amd import synthetic


@pytest.mark.parametrize("seed", [0, 1, 23111963, 23111963 + 7, 2**32 - 1, 2**32 + 5, 2**63 + 12345])
def test_oracle_seed_expansion_matches_numpy(seed):
    st = np.random.PCG64(seed).state["state"]
    assert src.seed_state(seed) == (st["state"], st["inc"])


@pytest.mark.parametrize("seed", [0, 23111963, 2**40 + 3])
def test_abi_seed_expansion_matches_numpy(seed):
    from tardis_amd import _lib
    L = _lib.lib()
    out = (C.c_uint64 * 4)()
    assert L.tardis_mc_pcg64_seed(seed, out) == 0
    st = np.random.PCG64(seed).state["state"]
    assert (out[0] << 64) | out[1] == st["state"]
    assert (out[2] << 64) | out[3] == st["inc"]


@pytest.mark.parametrize("n,max_val", [(1, 2**32 - 1), (257, 2**32 - 1), (1000, 4252017623), (1001, 3_000_000_000), (64, 10)])
def test_oracle_streams_match_numpy(n, max_val):
    seed = 23111963 + n
    rng = np.random.default_rng(seed)
    ref_seeds = rng.choice(max_val, n, replace=True)
    ref_xis = rng.random((5, n))
    ref_z = rng.random(n)
    seeds, xis, z = src.black_body_draws(seed, n, max_val)
    assert np.array_equal(seeds, ref_seeds)
    assert np.array_equal(xis, ref_xis)
    assert np.array_equal(z, ref_z)


def _host_packets(n, radius, temperature, seed_offset=0):
    return synthetic.black_body_packets(n, radius, temperature, seed_offset=seed_offset)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed_offset", [(1, 0), (1000, 3), (300_001, 1)])
def test_device_source_matches_host_run(n, seed_offset, oracle):
    from tardis_amd.engine import Engine
    eng = Engine(0)
    radius, T = 1.2e15, 9974.0
    ref = _host_packets(n, radius, T, seed_offset)
    eng.create_blackbody_packets(n, radius, T, seed_offset=seed_offset)
    got = eng.get_packets()
    assert np.array_equal(got["packet_seeds"], ref.packet_seeds)                 # bit-exact
    assert np.array_equal(got["initial_mus"], ref.initial_mus)                   # bit-exact (sqrt is IEEE)
    assert np.array_equal(got["initial_radii"], ref.initial_radii)
    assert np.array_equal(got["initial_energies"], ref.initial_energies)
    # nus: same draws, same operation order; bit-exact when the host evaluates the logarithm with the oracle's portable
    # log (the device's), and within a few ulp of numpy's own log (whose rounding depends on the host's SIMD dispatch;
    # the reference evaluates it with numexpr, black_body.py:182)
    rng = np.random.default_rng(synthetic.DEFAULT_BASE_SEED + seed_offset)
    rng.choice(synthetic.MAX_SEED_VAL, n, replace=True)
    xis = rng.random((5, n))
    l_array = np.cumsum(np.arange(1, 1000, dtype=np.float64) ** -4)
    l_min = l_array.searchsorted(xis[0] * (np.pi**4 / 90.0)) + 1.0
    x = -oracle.log_array(np.prod(xis[1:], 0), 1) / l_min
    from tardis_amd import state as st
    assert np.array_equal(got["initial_nus"], x * (st.K_BOLTZMANN * T) / st.H_PLANCK)
    np.testing.assert_allclose(got["initial_nus"], ref.initial_nus, rtol=2e-15, atol=0)
    # shards reproduce slices of the global draw
    if n > 10:
        lo, hi = n // 3, n // 3 + n // 2
        eng.create_blackbody_packets(n, radius, T, seed_offset=seed_offset, first=lo, count=hi - lo)
        part = eng.get_packets()
        for k in got:
            assert np.array_equal(part[k], got[k][lo:hi]), k
    eng.close()


@pytest.mark.gpu
def test_device_source_rejection_path():
    """A seed range with a 1 % Lemire rejection rate exercises the rejected-draw bookkeeping against numpy."""
    from tardis_amd.engine import Engine
    eng = Engine(0)
    n, max_val, seed = 20_000, 4252017623, 23111963
    rng = np.random.default_rng(seed)
    ref_seeds = rng.choice(max_val, n, replace=True)
    ref_xis = rng.random((5, n))
    ref_mu = np.sqrt(rng.random(n))
    eng.create_blackbody_packets(n, 1.0, 1.0e4, max_seed_val=max_val)
    got = eng.get_packets()
    assert np.array_equal(got["packet_seeds"], ref_seeds)
    assert np.array_equal(got["initial_mus"], ref_mu)       # the xi / mu streams start after the consumed u32 draws
    eng.close()


@pytest.mark.gpu
def test_device_source_feeds_transport_identically():
    """Packets sampled on the device propagate to the same result as the same packets handed in from the host."""
    from tardis_amd.engine import Engine
    prob = synthetic.make_problem(seed=5, n_packets=50_000, n_shells=20, n_lines=20_000)
    pc = prob.packet_collection
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion)
    eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
    eng.create_blackbody_packets(pc.number_of_packets, prob.geometry.r_inner[0], 1.0e4)
    dev = eng.get_packets()
    assert np.array_equal(dev["packet_seeds"], pc.packet_seeds)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    a = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    from tardis_amd import state as st
    pc2 = st.PacketCollection(dev["initial_radii"], dev["initial_nus"], dev["initial_mus"], dev["initial_energies"],
                              dev["packet_seeds"], pc.radiation_field_luminosity)
    eng.set_packets(pc2)
    eng.reset_estimators(); eng.propagate(); eng.synchronize()
    b = eng.get_results(track_last_interaction=False, want_line_estimators=False)
    assert np.array_equal(a.output_nus, b.output_nus) and np.array_equal(a.output_energies, b.output_energies)
