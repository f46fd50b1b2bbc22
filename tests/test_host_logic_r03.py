"""CPU tests of the host-side pieces added in round 3: the heavy-tailed block generator, the byte model of bench.py, and the
freshness rules of the resident result views (no GPU: a stand-in engine)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from tardis_amd import synthetic, transport

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_heavy_tailed_blocks_of_the_generator(oracle):
    for mode, rows in (("macroatom", 3), ("downbranch", 1)):
        prob = synthetic.make_problem(seed=5, n_packets=300, n_shells=6, n_lines=20_000, line_interaction_type=mode, level_sizes="heavy")
        op = prob.opacity_state
        edge = np.asarray(op.macro_block_edge_index)
        sizes = np.diff(edge)
        assert edge[0] == 0 and edge[-1] == len(op.transition_type) == rows * 20_000 and np.all(sizes > 0)
        for g in (11, 32, 33, 64, 100, 700, 6000):  # the planted sizes (6000 lines need 12 000 to be left)
            assert (sizes == g * rows).any(), g
        assert np.median(sizes) <= 8 * rows * 2 and sizes.max() == 6000 * rows
        # normalised per block and shell; most rows of the long blocks are below 2**-16 of their block's sum
        tp = op.transition_probabilities
        sums = np.add.reduceat(tp, edge[:-1], axis=0)
        np.testing.assert_allclose(sums, 1.0, rtol=1e-12)
        big = int(np.argmax(sizes))
        blk = tp[edge[big]:edge[big + 1], 0]
        assert (blk < 2.0**-16).mean() > 0.5 and blk.max() > 0.01
        # every line belongs to the block of its level; destinations are levels
        assert np.all(op.line2macro_level_upper >= 0) and op.line2macro_level_upper.max() == len(sizes) - 1
        internal = op.transition_type >= 0
        assert np.all(op.destination_level_id[internal] < len(sizes)) and np.all(op.destination_level_id[~internal] == -99)
        # the reference's serial walk examines far more rows per jump than any 4-8-line level holds
        ref = oracle.run(prob.packet_collection, prob.geometry, prob.time_explosion, op, prob.montecarlo_configuration,
                         prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, track_last_interaction=False)
        assert ref.return_code == 0
        jumps_upper = ref.counters["rng_draws"] - ref.counters["events"]
        assert ref.counters["macro_transitions"] > 30 * max(jumps_upper, 1) * (1 if mode == "macroatom" else 0.2)
    # the default generator is untouched by the option (fixtures and earlier rounds' numbers depend on it)
    a = synthetic.make_problem(seed=3, n_packets=10, n_shells=4, n_lines=500, line_interaction_type="macroatom")
    b = synthetic.make_problem(seed=3, n_packets=10, n_shells=4, n_lines=500, line_interaction_type="macroatom", level_sizes="uniform")
    assert np.array_equal(a.opacity_state.transition_probabilities, b.opacity_state.transition_probabilities)
    with pytest.raises(ValueError):
        synthetic.make_problem(seed=3, n_packets=1, n_lines=100, line_interaction_type="macroatom", level_sizes="nonsense")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_byte_model_of_the_bench():
    bench = _bench_module()
    # the macro-atom term: min(8 B x rows the reference's serial walk examines, 64 B x jumps) -- a jump needs at least one
    # 64-byte sector (round 4: the block's hot sector); J = rng_draws - 2 events is a lower bound of the jumps
    c = dict(line_visits=3278, events=91, macro_transitions=2385, vpacket_line_visits=0, vpackets=0, rng_draws=420, packets=1)
    assert bench.walk_bytes(c) == min(8.0 * 2385, 64.0 * (420 - 2 * 91)) == 64.0 * 238
    assert bench.walk_bytes(dict(c, macro_transitions=1500)) == 8.0 * 1500  # (short blocks: the serial walk is the cheaper bound)
    assert bench.algorithmic_bytes(c, "step") == 48 * 3278 + 56 * 91 + bench.walk_bytes(c) + 56
    assert bench.algorithmic_bytes(c, "propagate") + bench.algorithmic_bytes(c, "estimators") == bench.algorithmic_bytes(c, "step")
    # heavy-tailed blocks: the serial count is no lower bound of a search
    h = dict(c, macro_transitions=88_000)
    assert bench.walk_bytes(h) == 64.0 * (420 - 182)
    # v-packets: draws are not jumps (no cap), and with the screening the line visits of the v-packets are not bytes anyone moves
    v = dict(c, vpackets=812, vpacket_line_visits=225_000, rng_draws=2066)
    assert bench.walk_bytes(v) == 8.0 * 2385
    assert bench.algorithmic_bytes(v, "step") - bench.algorithmic_bytes(v, "step", screened=True) == 16.0 * 225_000


class _FakeEngine:
    def __init__(self):
        self.packets_generation = self.results_generation = 0
        self.n_drawn = None

    def create_blackbody_packets(self, n, radius, temperature, base_seed=0, seed_offset=0):
        self.packets_generation += 1
        self.n_drawn = (n, radius, temperature, base_seed, seed_offset)

    def get_packets(self):
        n = self.n_drawn[0]
        return {k: np.full(n, float(self.packets_generation)) for k in ("initial_radii", "initial_nus", "initial_mus", "initial_energies")} | {
            "packet_seeds": np.arange(n)}

    def get_results(self, **kw):
        import types
        n = self.n_drawn[0]
        return types.SimpleNamespace(output_nus=np.full(n, 1.0 + self.results_generation), output_energies=np.full(n, 0.5))


def test_resident_views_refuse_stale_data():
    eng = _FakeEngine()
    pc = transport.DevicePacketCollection(eng, 7, 1.2e15, 1.0e4, 23111963, 3)
    assert pc.number_of_packets == 7 and pc.time_of_simulation == 1 / pc.radiation_field_luminosity
    with pytest.raises(RuntimeError):
        pc.initial_nus  # nothing drawn yet
    pc._draw()
    assert eng.n_drawn == (7, 1.2e15, 1.0e4, 23111963, 3)
    assert np.all(pc.initial_nus == 1.0) and len(pc.packet_seeds) == 7
    with pytest.raises(RuntimeError):
        pc.output_nus  # not propagated yet
    eng.results_generation += 1
    pc._mark_propagated()
    first = pc.output_nus
    assert np.all(first == 2.0) and pc.output_nus is first  # fetched once
    # the engine runs something else: un-read results of this collection are gone, and say so
    pc2 = transport.DevicePacketCollection(eng, 7, 1.2e15, 1.0e4, 23111963, 4)
    pc2._draw()
    eng.results_generation += 1
    pc2._mark_propagated()
    assert np.all(pc.output_nus == 2.0)          # (already on the host: still this run's)
    pc_unread = transport.DevicePacketCollection(eng, 7, 1.2e15, 1.0e4, 23111963, 5)
    pc_unread._draw(); eng.results_generation += 1; pc_unread._mark_propagated()
    with pytest.raises(RuntimeError):
        pc2.output_nus  # never read before the engine moved on


def test_resident_solver_argument_checks():
    grid = np.linspace(1e14, 1e15, 11)
    solver = transport.MCTransportSolverHIP(grid, resident=False)
    with pytest.raises(ValueError):
        solver.initialize_transport_state(None, synthetic.make_geometry(3), None, 1.0e6, n_packets=10, temperature_inner=1e4)
    ts = transport.MonteCarloTransportState(None, None, None, 1.0)
    with pytest.raises(RuntimeError):
        ts.radiation_field(np.ones(3))
