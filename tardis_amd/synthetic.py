"""Deterministic synthetic transport problems shaped like BASELINE.json's configs.

The reference's atomic data (kurucz_cd23_chianti_H_He) is fetched over the network and is not available
offline, so ``tardis_example.yml`` cannot be run literally.  ``make_problem`` builds opacity states with the
same *layout and statistics* the classic transport mode sees (sorted-descending line list, [L,S] Sobolev
optical depths, macro-atom block tables) on the tardis_example geometry (20 shells, 1.1e9-2.0e9 cm/s,
t_exp = 13 d; docs/tardis_example.yml:6,12-16).

The packet source is a host-side restatement of the reference's ``BlackBodySimpleSource``
(tardis/transport/montecarlo/packet_source/base.py:195-253, black_body.py:122-222): PCG64 stream seeded
with ``base_seed + iteration``; draw order = packet seeds, 5xP uniforms for the Carter-Cashwell Planck
sampler, P uniforms for mu = sqrt(z).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import state as st

DAY = 86400.0
ANGSTROM = 1e-8
DEFAULT_BASE_SEED = 23111963  # io/configuration/schemas/montecarlo.yml:12-15
MAX_SEED_VAL = 2**32 - 1  # packet_source/base.py:25


@dataclass
class Problem:
    packet_collection: st.PacketCollection
    geometry: st.HomologousRadial1DGeometry
    time_explosion: float
    opacity_state: st.OpacityState
    montecarlo_configuration: st.MonteCarloConfiguration
    spectrum_frequency_grid: np.ndarray
    description: str = ""


def black_body_packets(n_packets: int, radius: float, temperature: float, base_seed: int = DEFAULT_BASE_SEED,
                       seed_offset: int = 0, l_samples: int = 1000, legacy_random_state=None) -> st.PacketCollection:
    """Sample a packet collection at the photosphere (see module docstring for the reference lines).

    ``legacy_random_state``: the reference's ``legacy_mode_enabled`` source (what its own integration test runs,
    tests/test_montecarlo_main_loop.py:14-60) draws the five Planck uniforms and the direction uniforms from NumPy's GLOBAL
    legacy stream (black_body.py:172,201), seeded once with ``base_seed`` when the source is constructed (base.py:48-59) and
    never re-seeded, so consecutive iterations continue it; only the packet seeds still come from PCG64(base_seed +
    iteration).  Pass one ``np.random.RandomState(base_seed)`` for the whole run to reproduce that."""
    rng = np.random.default_rng(base_seed + seed_offset)
    packet_seeds = rng.choice(MAX_SEED_VAL, n_packets, replace=True)
    radii = np.ones(n_packets) * radius
    # Planck sampler (Carter & Cashwell 1975 via Bjorkman & Wood 2001)
    l_array = np.cumsum(np.arange(1, l_samples, dtype=np.float64) ** -4)
    l_coef = np.pi**4 / 90.0
    draws = legacy_random_state.random_sample if legacy_random_state is not None else rng.random
    xis = draws((5, n_packets))
    l_min = l_array.searchsorted(xis[0] * l_coef) + 1.0
    x = -np.log(np.prod(xis[1:], 0)) / l_min
    nus = x * (st.K_BOLTZMANN * temperature) / st.H_PLANCK
    mus = np.sqrt(draws(n_packets))
    energies = np.ones(n_packets) / n_packets
    luminosity = 4 * np.pi * st.SIGMA_SB * radius**2 * temperature**4
    return st.PacketCollection(radii, nus, mus, energies, packet_seeds, luminosity)


def make_geometry(n_shells: int = 20, v_inner: float = 1.1e9, v_outer: float = 2.0e9,
                  time_explosion: float = 13 * DAY) -> st.HomologousRadial1DGeometry:
    v = np.linspace(v_inner, v_outer, n_shells + 1)
    r = v * time_explosion
    return st.HomologousRadial1DGeometry(r[:-1], r[1:], v[:-1], v[1:], time_explosion)


def make_opacity_state(seed: int, geometry: st.HomologousRadial1DGeometry, n_lines: int,
                       line_interaction_type: str, log_tau_mean: float = -4.0, log_tau_sigma: float = 2.0,
                       electron_density_0: float = 1e9,
                       shell_independent_probabilities: bool = False,
                       level_sizes: str = "uniform") -> st.OpacityState:
    rng = np.random.default_rng(seed)
    n_shells = len(geometry.r_inner)
    # lines: log-uniform in wavelength on [500 A, 20000 A]; frequency sorted descending
    lam = np.exp(rng.uniform(np.log(500.0), np.log(20000.0), n_lines)) * ANGSTROM
    line_list_nu = np.sort(st.C_SPEED_OF_LIGHT / lam)[::-1].copy()
    rho = (geometry.v_inner / geometry.v_inner[0]) ** -7.0
    tau0 = 10.0 ** rng.normal(log_tau_mean, log_tau_sigma, n_lines)
    tau_sobolev = tau0[:, None] * rho[None, :]
    electron_density = electron_density_0 * rho
    t_electrons = np.full(n_shells, 9000.0)

    if line_interaction_type == "scatter":
        # size-1 dummies, as opacity_state.py:199-209 does for scatter mode
        op = st.OpacityState(electron_density, t_electrons, line_list_nu, tau_sobolev,
                             np.zeros((1, n_shells)), np.zeros(1, np.int64), np.zeros(1, np.int64),
                             np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64))
        op.tau_factors = (tau0, rho)
        return op

    # macro-atom levels: every level owns 4-8 emission lines ("uniform"), lines assigned at random so that a
    # de-excitation can fluoresce to a far-away wavelength
    sizes = []
    left = n_lines
    if level_sizes == "heavy":
        # In the reference a block is ALL transitions out of one source level (macroatom_solver.py:383-428, 624-670):
        # with real Kurucz data most levels own a handful of lines, Fe-group levels hundreds to thousands.  Pareto(1.2)
        # line counts from 4 up (2 % of the levels > 100 lines, 0.1 % > 1000), capped at 6000 lines per level, behind a few
        # planted sizes that pin the edges of the device tables: 11 / 33 lines (33 / 99 rows in macroatom mode, 33 rows
        # in downbranch: just past one 32-entry window, not a multiple of 8), 32 and 64 (whole windows: no padding),
        # 100, 700 and -- on long line lists -- 6000 lines.
        for g in (11, 32, 33, 64, 100, 700, 6000):
            if left >= 2 * g:
                sizes.append(g)
                left -= g
        while left > 0:
            g = int(min(left, 6000, np.floor(4.0 * (1.0 - rng.random()) ** (-1.0 / 1.2))))
            sizes.append(g)
            left -= g
        order = rng.permutation(len(sizes))  # (planted blocks somewhere in the tables, not all at the front)
        sizes = [sizes[i] for i in order]
    elif level_sizes == "uniform":
        while left > 0:
            g = int(min(left, rng.integers(4, 9)))
            sizes.append(g)
            left -= g
    else:
        raise ValueError(level_sizes)
    sizes = np.asarray(sizes)
    n_levels = len(sizes)
    perm = rng.permutation(n_lines)
    level_of_line = np.empty(n_lines, np.int64)
    level_lines = []
    pos = 0
    for lvl, g in enumerate(sizes):
        ids = np.sort(perm[pos:pos + g])
        level_lines.append(ids)
        level_of_line[ids] = lvl
        pos += g

    if line_interaction_type == "downbranch":
        rows_per_line = 1
    elif line_interaction_type == "macroatom":
        rows_per_line = 3
    else:
        raise ValueError(line_interaction_type)
    n_trans = rows_per_line * n_lines
    transition_type = np.empty(n_trans, np.int64)
    destination_level_id = np.empty(n_trans, np.int64)
    transition_line_id = np.empty(n_trans, np.int64)
    block_edge = np.empty(n_levels + 1, np.int64)
    weights = rng.random((n_trans, 1 if shell_independent_probabilities else n_shells)) + 0.05
    if level_sizes == "heavy":
        # transition probabilities of a real block span many decades (A-values x Sobolev escape probabilities): most
        # rows of a long block are below 2**-16 of the block's sum, a few carry it
        weights *= 10.0 ** rng.normal(0.0, 2.0, (n_trans, 1))
    row = 0
    for lvl, ids in enumerate(level_lines):
        block_edge[lvl] = row
        g = len(ids)
        # emission rows (BB_EMISSION = -1; destination unused = -99 as in the reference)
        transition_type[row:row + g] = -1
        destination_level_id[row:row + g] = -99
        transition_line_id[row:row + g] = ids
        row += g
        if rows_per_line == 3:
            # internal down (0) and internal up (1) jumps to random other levels; emission rows keep
            # >= ~1/3 of the block probability so the walk terminates quickly
            for ttype in (0, 1):
                transition_type[row:row + g] = ttype
                destination_level_id[row:row + g] = rng.integers(0, n_levels, g)
                transition_line_id[row:row + g] = ids
                row += g
    block_edge[n_levels] = n_trans  # trailing sentinel (macroatom_solver.py:651-656)
    # normalise per block per shell
    for lvl in range(n_levels):
        a, b = block_edge[lvl], block_edge[lvl + 1]
        weights[a:b] /= weights[a:b].sum(axis=0, keepdims=True)
    if shell_independent_probabilities:
        weights = np.repeat(weights, n_shells, axis=1)
    op = st.OpacityState(electron_density, t_electrons, line_list_nu, tau_sobolev, weights, level_of_line,
                         block_edge, transition_type, destination_level_id, transition_line_id)
    op.tau_factors = (tau0, rho)  # tau_sobolev == tau0[:, None] * rho[None, :] exactly (compact fixtures)
    return op


def make_spectrum_grid(n_bins: int = 10000, lam_start: float = 500.0, lam_stop: float = 20000.0) -> np.ndarray:
    """Uniform-in-frequency ascending edge grid (spectrum/base.py:190-195; solver.py:305-309)."""
    return np.linspace(st.C_SPEED_OF_LIGHT / (lam_stop * ANGSTROM), st.C_SPEED_OF_LIGHT / (lam_start * ANGSTROM),
                       n_bins + 1)


def make_problem(seed: int = 1, n_packets: int = 10_000, n_shells: int = 20, n_lines: int = 30_000,
                 line_interaction_type: str = "downbranch", n_vpackets: int = 0,
                 enable_full_relativity: bool = False, disable_line_scattering: bool = False,
                 n_bins: int = 10_000, iteration: int = 0, temperature_inner: float = 1.0e4,
                 electron_density_0: float = 1e9, log_tau_mean: float = -4.0,
                 vpacket_spawn_range=None, shell_independent_probabilities: bool = False,
                 level_sizes: str = "uniform") -> Problem:
    geometry = make_geometry(n_shells)
    opacity = make_opacity_state(seed, geometry, n_lines, line_interaction_type,
                                 log_tau_mean=log_tau_mean, electron_density_0=electron_density_0,
                                 shell_independent_probabilities=shell_independent_probabilities,
                                 level_sizes=level_sizes)
    packets = black_body_packets(n_packets, geometry.r_inner[0], temperature_inner, seed_offset=iteration)
    cfg = st.MonteCarloConfiguration()
    cfg.LINE_INTERACTION_TYPE = st.LINE_INTERACTION_TYPES[line_interaction_type]
    cfg.NUMBER_OF_VPACKETS = n_vpackets
    cfg.TEMPORARY_V_PACKET_BINS = n_vpackets
    cfg.ENABLE_FULL_RELATIVITY = bool(enable_full_relativity)
    cfg.DISABLE_LINE_SCATTERING = bool(disable_line_scattering)
    cfg.MONTECARLO_SEED = DEFAULT_BASE_SEED
    if vpacket_spawn_range is not None:
        cfg.VPACKET_SPAWN_START_FREQUENCY, cfg.VPACKET_SPAWN_END_FREQUENCY = vpacket_spawn_range
    grid = make_spectrum_grid(n_bins)
    desc = (f"synthetic P={n_packets} S={n_shells} L={n_lines} {line_interaction_type} n_v={n_vpackets} "
            f"full_rel={int(enable_full_relativity)} seed={seed}" + (" heavy-tailed levels" if level_sizes == "heavy" else ""))
    return Problem(packets, geometry, geometry.time_explosion, opacity, cfg, grid, desc)


# BASELINE.json configs -> generator arguments
BASELINE_CONFIGS = {
    1: dict(n_packets=10_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=0),
    2: dict(n_packets=10_000_000, n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=0),
    3: dict(n_packets=100_000_000, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=0),
    4: dict(n_packets=100_000_000, n_shells=20, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=0),
    5: dict(n_packets=500_000_000, n_shells=100, n_lines=500_000, line_interaction_type="macroatom",
            n_vpackets=10),
}
