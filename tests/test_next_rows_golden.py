"""The rows next to the hot path against the REFERENCE's own code.

tests/golden/packet_source_*.npz and radfield_*.npz hold what the reference's BlackBodySimpleSource.create_packets
(tardis/transport/montecarlo/packet_source/base.py:195-253, black_body.py:140-222) and
MCRadiationFieldPropertiesSolver.solve (estimators/mc_rad_field_solver.py:37-144) produced in the dev container
(tools/make_golden_next_rows.py imports them unmodified).  Held to them here:
  * CPU: the oracle restatements (oracle/pcg64_source.py, oracle/radfield.py) and the host sampler synthetic.black_body_packets;
  * GPU: the device packet source and the device radiation-field kernels.
Integers and square roots are exact; frequencies go through a logarithm (the reference's is numexpr's, the fixtures' numpy's,
the device's the portable one: all within 2 ulp) -> rtol 1e-15..1e-14; the radiation field through exp -> rtol 1e-13.
"""
import ast
import glob
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import pcg64_source, radfield
from tardis_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SOURCE_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "packet_source_*.npz")))
RADFIELD_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "radfield_*.npz")))


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def test_fixtures_exist():
    assert len(SOURCE_CASES) >= 4 and len(RADFIELD_CASES) >= 3


@pytest.mark.parametrize("name", SOURCE_CASES)
def test_host_sampler_and_oracle_streams_match_the_reference_source(name):
    g = _load(name)
    n = int(g["n"])
    pc = synthetic.black_body_packets(n, float(g["radius"]), float(g["temperature"]), base_seed=int(g["base_seed"]),
                                      seed_offset=int(g["seed_offset"]))
    assert np.array_equal(pc.packet_seeds, g["packet_seeds"])
    assert np.array_equal(pc.initial_mus, g["initial_mus"])
    assert np.array_equal(pc.initial_radii, g["initial_radii"])
    assert np.array_equal(pc.initial_energies, g["initial_energies"])
    assert_allclose(pc.initial_nus, g["initial_nus"], rtol=1e-15, atol=0)
    assert_allclose(pc.radiation_field_luminosity, float(g["radiation_field_luminosity"]), rtol=1e-15)
    # the plain-integer restatement of NumPy's PCG64 streams (the device source's checker)
    seeds, xis, z = pcg64_source.black_body_draws(int(g["base_seed"]) + int(g["seed_offset"]), n, 2**32 - 1)
    assert np.array_equal(seeds, g["packet_seeds"])
    assert np.array_equal(np.sqrt(z), g["initial_mus"])


def test_host_sampler_legacy_mode_matches_the_reference_source():
    """legacy_mode_enabled (what the reference's own integration test uses, tests/test_montecarlo_main_loop.py:14-60): Planck
    and direction uniforms from the global legacy MT19937 stream, seeded once, continued over two iterations."""
    state = None
    for it in (0, 1):
        g = _load(f"legacy_packet_source_iter{it}")
        if state is None:
            state = np.random.RandomState(int(g["base_seed"]))
        pc = synthetic.black_body_packets(int(g["n"]), float(g["radius"]), float(g["temperature"]), base_seed=int(g["base_seed"]),
                                          seed_offset=it, legacy_random_state=state)
        assert np.array_equal(pc.packet_seeds, g["packet_seeds"])
        assert np.array_equal(pc.initial_mus, g["initial_mus"])
        assert_allclose(pc.initial_nus, g["initial_nus"], rtol=1e-15, atol=0)
    # and the legacy stream really is another stream than the PCG64 one
    assert not np.array_equal(synthetic.black_body_packets(777, 1.2e15, 9974.0).initial_mus, _load("legacy_packet_source_iter0")["initial_mus"])


@pytest.mark.parametrize("name", RADFIELD_CASES)
def test_radfield_oracle_matches_the_reference_solver(name):
    g = _load(name)
    t_rad, w, jb = radfield.solve(g["in_j_estimator"], g["in_nu_bar_estimator"], g["in_j_blue_estimator"].copy(),
                                  float(g["in_time_explosion"]), float(g["in_time_of_simulation"]), g["in_volume"],
                                  g["in_line_list_nu"], w_epsilon=float(g["w_epsilon"]),
                                  detailed_optical_window=bool(g["detailed_optical_window"]))
    assert_allclose(t_rad, g["t_radiative"], rtol=1e-14, atol=0)
    assert_allclose(w, g["dilution_factor"], rtol=1e-13, atol=0)
    assert_allclose(jb, g["j_blues"], rtol=1e-13, atol=0)
    assert (g["in_j_blue_estimator"] == 0).any()  # the dilute-Planck fill-in branch is exercised


@pytest.mark.gpu
@pytest.mark.parametrize("name", SOURCE_CASES)
def test_device_source_matches_the_reference_source(name):
    from tardis_amd.engine import Engine
    g = _load(name)
    n = int(g["n"])
    with Engine(0) as eng:
        eng.create_blackbody_packets(n, float(g["radius"]), float(g["temperature"]), base_seed=int(g["base_seed"]),
                                     seed_offset=int(g["seed_offset"]))
        got = eng.get_packets()
        # a shard of the same draw (what a rank of a multi-GPU job generates)
        if n > 2:
            lo, hi = n // 3, n - 1
            eng.create_blackbody_packets(n, float(g["radius"]), float(g["temperature"]), base_seed=int(g["base_seed"]),
                                         seed_offset=int(g["seed_offset"]), first=lo, count=hi - lo)
            part = eng.get_packets()
            for k in ("initial_nus", "initial_mus", "packet_seeds"):
                assert np.array_equal(part[k], got[k][lo:hi]), k
    assert np.array_equal(got["packet_seeds"], g["packet_seeds"])
    assert np.array_equal(got["initial_mus"], g["initial_mus"])
    assert np.array_equal(got["initial_radii"], g["initial_radii"])
    assert np.array_equal(got["initial_energies"], g["initial_energies"])
    assert_allclose(got["initial_nus"], g["initial_nus"], rtol=4e-15, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RADFIELD_CASES)
def test_device_radiation_field_matches_the_reference_solver(name):
    """The engine runs the fixture's problem itself (its estimators agree with the oracle's, which the reference solver was
    fed, to the summation-order tolerance 1e-11) and updates the radiation field on the device."""
    from tardis_amd.engine import Engine
    g = _load(name)
    prob = synthetic.make_problem(**ast.literal_eval(str(g["problem_args"])))
    with Engine(0) as eng:
        eng.set_geometry(prob.geometry, prob.time_explosion)
        eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
        eng.set_packets(prob.packet_collection)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        res = eng.get_results(track_last_interaction=False)
        got = eng.radiation_field(float(g["in_time_of_simulation"]), g["in_volume"], w_epsilon=float(g["w_epsilon"]),
                                  detailed_optical_window=bool(g["detailed_optical_window"]))
    assert_allclose(res.j_estimator, g["in_j_estimator"], rtol=1e-11)
    assert np.array_equal(res.j_blue_estimator == 0, g["in_j_blue_estimator"] == 0)
    assert_allclose(got["t_radiative"], g["t_radiative"], rtol=1e-10, atol=0)
    assert_allclose(got["dilution_factor"], g["dilution_factor"], rtol=1e-10, atol=0)
    assert_allclose(got["j_blues"], g["j_blues"], rtol=1e-10, atol=0)
