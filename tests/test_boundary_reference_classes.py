"""The drop-in boundary executed with the REFERENCE'S OWN classes (build container only; CPU).

`tests/test_boundary_host.py` feeds the marshalling stand-ins; here the objects are the reference's: `PacketCollection`
(packets/packet_collections.py:14-76), `OpacityStateNumba` (opacities/opacity_state_numba.py:14-196),
`NumbaHomologousRadial1DGeometry` (model/geometry/radial1d_homologous.py:199-226), `MonteCarloConfiguration`
(configuration/base.py:11-49) and a list of `TrackerLastInteraction` (packets/trackers/tracker_last_interaction.py:8-254),
imported from /root/reference through tools/ref_shim.py.  They go

  * through `_abi.marshal_*`: every struct field and every buffer equals what the `tardis_amd.state` containers give;
  * through `transport.montecarlo_transport_with_vpackets` as `run_classic` calls it (modes/classic/solver.py:223-234), with
    the C library replaced by a checker that reads the marshalled C structs back and lets the CPU oracle compute the
    results: outputs written in place, the reference's tracker objects, estimators and v-packet log equal the fixture the
    reference itself produced;
  * and through the reference's own `MCTransportSolverClassic.run_classic` with INTEGRATION.md's one-line rebinding of
    `solver.montecarlo_transport_with_vpackets`: the reference's post-processing (`trackers_last_interaction_to_df`,
    estimator attachment, v-packet tracker) runs on what the wrapper returned.

Nothing of the reference travels: the module is skipped where /root/reference does not exist (the GPU box).
"""
import ctypes as C
import os
import sys
import types

import numpy as np
import pytest

from tardis_amd import _abi, _lib, state as st, transport
from tardis_amd.engine import Engine

import _golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TREE = os.environ.get("TARDIS_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_TREE, "tardis")),
                                reason="the reference tree is only present in the build container")

CASES = ["macroatom_nv3_log", "downbranch_nv2_roulette", "scatter_nv0", "macroatom_fullrel_nv2"]


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_shim

    return ref_shim.load()


def _reference_objects(ref, prob):
    """The reference's own containers around the fixture's inputs (constructor signatures of the cited files)."""
    pc, op, geo, cfg = prob.packet_collection, prob.opacity_state, prob.geometry, prob.montecarlo_configuration
    rpc = ref.PacketCollection(pc.initial_radii.copy(), pc.initial_nus.copy(), pc.initial_mus.copy(),
                               pc.initial_energies.copy(), pc.packet_seeds.copy(), pc.radiation_field_luminosity)
    rgeo = ref.NumbaHomologousRadial1DGeometry(geo.r_inner, geo.r_outer, geo.v_inner, geo.v_outer, geo.time_explosion)
    rop = ref.OpacityStateNumba(
        op.electron_density, op.t_electrons, op.line_list_nu, op.tau_sobolev, op.transition_probabilities,
        op.line2macro_level_upper, op.macro_block_edge_index, op.transition_type, op.destination_level_id,
        op.transition_line_id, np.zeros(0), np.zeros((0, 0)), np.zeros(0), np.zeros(0), np.zeros(0, np.int64),
        np.zeros((0, 0)), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros((0, 0)), np.zeros(0, np.int64), -1)
    rcfg = ref.MonteCarloConfiguration()
    for k, v in vars(cfg).items():
        setattr(rcfg, k, v)
    return rpc, rgeo, rop, rcfg


def _ptr_array(p, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)


def _opacity_buffers(s):
    L, S, T, E = s.n_lines, s.n_shells, s.n_transitions, s.n_macro_block_edges
    return dict(n=(L, S, T, E), electron_density=_ptr_array(s.electron_density, S, np.float64),
                line_list_nu=_ptr_array(s.line_list_nu, L, np.float64), tau_sobolev=_ptr_array(s.tau_sobolev, L * S, np.float64),
                transition_probabilities=_ptr_array(s.transition_probabilities, T * S, np.float64),
                line2macro_level_upper=_ptr_array(s.line2macro_level_upper, L, np.int64),
                macro_block_edge_index=_ptr_array(s.macro_block_edge_index, E, np.int64),
                transition_type=_ptr_array(s.transition_type, T, np.int64),
                destination_level_id=_ptr_array(s.destination_level_id, T, np.int64),
                transition_line_id=_ptr_array(s.transition_line_id, T, np.int64))


def _packet_buffers(s):
    n = s.n_packets
    return dict(n=n, **{k: _ptr_array(getattr(s, k), n, np.float64) for k in ("initial_radii", "initial_nus", "initial_mus", "initial_energies")},
                packet_seeds=_ptr_array(s.packet_seeds, n, np.int64))


def _geometry_buffers(s):
    n = s.n_shells
    return dict(n=n, r_inner=_ptr_array(s.r_inner, n, np.float64), r_outer=_ptr_array(s.r_outer, n, np.float64), t=s.time_explosion)


def _config_values(s):
    d = {name: getattr(s, name) for name, _ in s._fields_ if name != "spectrum_frequency_grid"}
    d["grid"] = _ptr_array(s.spectrum_frequency_grid, s.n_spectrum_grid, np.float64)
    return d


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


@pytest.mark.parametrize("name", CASES)
def test_marshalling_of_the_reference_classes_equals_the_state_containers(ref, name):
    prob, g = _golden.load_case(name)
    rpc, rgeo, rop, rcfg = _reference_objects(ref, prob)
    assert type(rpc).__module__.startswith("tardis.") and type(rop).__module__.startswith("tardis.")
    _same(_packet_buffers(_abi.marshal_packets(rpc).struct), _packet_buffers(_abi.marshal_packets(prob.packet_collection).struct))
    _same(_geometry_buffers(_abi.marshal_geometry(rgeo, prob.time_explosion).struct),
          _geometry_buffers(_abi.marshal_geometry(prob.geometry, prob.time_explosion).struct))
    # (the reference's geometry carries time_explosion itself, too)
    assert _abi.marshal_geometry(rgeo).struct.time_explosion == prob.time_explosion
    _same(_opacity_buffers(_abi.marshal_opacity(rop).struct), _opacity_buffers(_abi.marshal_opacity(prob.opacity_state).struct))
    n_v = int(rcfg.NUMBER_OF_VPACKETS)
    _same(_config_values(_abi.marshal_config(rcfg, prob.spectrum_frequency_grid, n_v).struct),
          _config_values(_abi.marshal_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid, n_v).struct))
    # and the buffers are the fixture's inputs in the header's layout ([n_lines, n_shells] row-major)
    ob = _opacity_buffers(_abi.marshal_opacity(rop).struct)
    assert np.array_equal(ob["tau_sobolev"].reshape(prob.opacity_state.tau_sobolev.shape), prob.opacity_state.tau_sobolev)
    assert ob["n"][:2] == prob.opacity_state.tau_sobolev.shape


class _CheckerLibrary:
    """Stands where libtardis_mc_hip.so stands (tardis_amd._lib.lib()): reads the C structs the boundary marshals back into arrays,
    lets the CPU oracle compute the results (test infrastructure: oracle/), and writes them through the TardisMcResult pointers.
    Records every call, so the test sees what reached the C ABI."""

    def __init__(self, oracle):
        self.oracle = oracle
        self.calls = []
        self.options = {}
        self.geo = self.op = self.cfg = self.pk = None
        self.result = None

    # -- lifetime / misc
    def tardis_mc_create(self, device, h):
        self.calls.append("create")
        return 0

    def tardis_mc_destroy(self, h):
        self.calls.append("destroy")

    def tardis_mc_last_error(self, h):
        return b""

    def tardis_mc_set_option(self, h, name, value):
        self.options[name.decode()] = int(value)
        return 0

    # -- staged inputs (the byref() objects of Marshalled.ref())
    def tardis_mc_set_geometry(self, h, ref_):
        self.calls.append("set_geometry"); self.geo = _geometry_buffers(ref_._obj); return 0

    def tardis_mc_set_opacity(self, h, ref_):
        self.calls.append("set_opacity"); self.op = _opacity_buffers(ref_._obj); return 0

    def tardis_mc_set_config(self, h, ref_):
        self.calls.append("set_config"); self.cfg = _config_values(ref_._obj); return 0

    def tardis_mc_set_packets(self, h, ref_):
        self.calls.append("set_packets"); self.pk = _packet_buffers(ref_._obj); return 0

    def tardis_mc_reset_estimators(self, h):
        self.calls.append("reset_estimators"); return 0

    def tardis_mc_stream_results(self, h, ref_):  # (optional: this stand-in fills the arrays in get_results)
        return 0

    def tardis_mc_synchronize(self, h):
        self.calls.append("synchronize"); return 0

    def tardis_mc_last_propagate_ms(self, h, out):
        out._obj.value = 0.0
        return 0

    def tardis_mc_progress(self, h, started, total):
        self.calls.append("progress")
        n = self.pk["n"] if self.pk else 0
        started._obj.value = n if self.result is not None else 0
        total._obj.value = n
        return 0

    def tardis_mc_propagate(self, h):
        self.calls.append("propagate")
        L, S, T, E = self.op["n"]
        op = st.OpacityState(self.op["electron_density"], np.zeros(S), self.op["line_list_nu"], self.op["tau_sobolev"].reshape(L, S),
                             self.op["transition_probabilities"].reshape(T, S), self.op["line2macro_level_upper"],
                             self.op["macro_block_edge_index"], self.op["transition_type"], self.op["destination_level_id"],
                             self.op["transition_line_id"])
        t = self.geo["t"]
        geo = st.HomologousRadial1DGeometry(self.geo["r_inner"], self.geo["r_outer"], self.geo["r_inner"] / t, self.geo["r_outer"] / t, t)
        pc = st.PacketCollection(self.pk["initial_radii"], self.pk["initial_nus"], self.pk["initial_mus"], self.pk["initial_energies"],
                                 self.pk["packet_seeds"], 1.0)
        c = self.cfg
        cfg = st.MonteCarloConfiguration()
        cfg.ENABLE_FULL_RELATIVITY = bool(c["enable_full_relativity"]); cfg.LINE_INTERACTION_TYPE = int(c["line_interaction_type"])
        cfg.DISABLE_LINE_SCATTERING = bool(c["disable_line_scattering"]); cfg.ENABLE_VPACKET_TRACKING = bool(c["enable_vpacket_tracking"])
        cfg.NUMBER_OF_VPACKETS = int(c["number_of_vpackets"]); cfg.SURVIVAL_PROBABILITY = c["survival_probability"]
        cfg.VPACKET_TAU_RUSSIAN = c["vpacket_tau_russian"]; cfg.VPACKET_SPAWN_START_FREQUENCY = c["vpacket_spawn_start_frequency"]
        cfg.VPACKET_SPAWN_END_FREQUENCY = c["vpacket_spawn_end_frequency"]
        assert c["sigma_thomson"] == st.SIGMA_THOMSON
        self.result = self.oracle.run(pc, geo, t, op, cfg, c["grid"], math_mode=self.oracle.MATH_LIBM,
                                      track_last_interaction=bool(self.options.get("track_last_interaction", 1)))
        return 0

    def tardis_mc_get_results(self, h, ref_):
        self.calls.append("get_results")
        r, o = ref_._obj, self.result

        def put(ptr, a):
            if ptr and a is not None:
                a = np.ascontiguousarray(a)
                C.memmove(ptr, a.ctypes.data, a.nbytes)

        put(r.output_nus, o.output_nus); put(r.output_energies, o.output_energies)
        put(r.j_estimator, o.j_estimator); put(r.nu_bar_estimator, o.nu_bar_estimator)
        put(r.j_blue_estimator, o.j_blue_estimator); put(r.edotlu_estimator, o.edotlu_estimator)
        put(r.v_packets_energy_hist, o.v_packets_energy_hist)
        if o.trackers is not None and r.li_radius:
            names = {"li_radius": "radius", "li_nu": "nu", "li_energy": "energy", "li_before_nu": "before_nu", "li_before_mu": "before_mu",
                     "li_before_energy": "before_energy", "li_after_nu": "after_nu", "li_after_mu": "after_mu", "li_after_energy": "after_energy",
                     "li_shell_id": "shell_id", "li_interaction_type": "interaction_type", "li_line_absorb_id": "interaction_line_absorb_id",
                     "li_line_emit_id": "interaction_line_emit_id", "li_interactions_count": "interactions_count"}
            for k, v in names.items():
                put(getattr(r, k), getattr(o.trackers, v))
        n_log = len(o.vpacket_nus) if getattr(o, "vpacket_nus", None) is not None else 0
        r.vpacket_log_count = n_log
        if n_log and r.vpacket_log_capacity >= n_log:
            put(r.vpacket_nus, o.vpacket_nus); put(r.vpacket_energies, o.vpacket_energies)
            put(r.vpacket_initial_mus, o.vpacket_initial_mus); put(r.vpacket_initial_rs, o.vpacket_initial_rs)
        r.first_error_packet = -1
        r.error_code = 0
        return 0


@pytest.fixture()
def checker_engine(oracle, monkeypatch):
    lib = _CheckerLibrary(oracle)
    monkeypatch.setattr(_lib, "_lib", lib)
    eng = Engine(0)
    yield eng, lib
    eng.close()


def _check_against_golden(g, rpc, trackers, hist, vt, eb, el, n_v, vlog):
    assert np.array_equal(rpc.output_nus, g["output_nus"]) and np.array_equal(rpc.output_energies, g["output_energies"])
    for f in _golden.TRACKER_F64:
        got = np.array([getattr(t, f) for t in trackers], dtype=np.float64)
        assert np.array_equal(got, g["trk_" + f], equal_nan=True), f
    for f in _golden.TRACKER_I64:
        got = np.array([getattr(t, f) for t in trackers], dtype=np.int64)
        assert np.array_equal(got, g["trk_" + f]), f
    # (the checker behind the C ABI is the serial CPU oracle: the reference's own summation order)
    assert np.array_equal(eb.mean_intensity_total, g["j_estimator"]) and np.array_equal(eb.mean_frequency, g["nu_bar_estimator"])
    stride = int(g["line_estimator_stride"])
    assert np.array_equal(el.mean_intensity_blueward[::stride], g["j_blue_estimator"])
    assert np.array_equal(el.energy_deposition_line_rate[::stride], g["edotlu_estimator"])
    assert np.array_equal(hist, g["v_packets_energy_hist"])
    if vlog:
        assert np.array_equal(vt.nus, g["vpacket_nus"]) and np.array_equal(vt.energies, g["vpacket_energies"])
        assert np.array_equal(vt.initial_mus, g["vpacket_initial_mus"]) and np.array_equal(vt.initial_rs, g["vpacket_initial_rs"])


@pytest.mark.parametrize("name", CASES)
def test_wrapper_called_like_run_classic_with_the_reference_objects(ref, checker_engine, name):
    """modes/classic/solver.py:223-234: the reference's objects straight through the wrapper; what the reference's own run of
    the same inputs produced (the fixture) comes back in the reference's objects."""
    eng, lib = checker_engine
    prob, g = _golden.load_case(name)
    rpc, rgeo, rop, rcfg = _reference_objects(ref, prob)
    n_v = int(rcfg.NUMBER_OF_VPACKETS)
    trackers = [ref.TrackerLastInteraction() for _ in range(rpc.number_of_packets)]
    hist, vt, eb, el = transport.montecarlo_transport_with_vpackets(
        rpc, rgeo, prob.time_explosion, rop, rcfg, prob.spectrum_frequency_grid, trackers, n_v,
        show_progress_bars=False, packet_propagation_function=ref.packet_propagation, engine=eng)
    assert lib.calls[:1] == ["create"]
    assert [c for c in lib.calls if c.startswith("set_")] == ["set_geometry", "set_opacity", "set_config", "set_packets"]
    assert lib.options["track_last_interaction"] == 1
    # what reached the C ABI is the fixture's input, in the header's layouts
    assert np.array_equal(lib.pk["packet_seeds"], prob.packet_collection.packet_seeds) and lib.pk["packet_seeds"].dtype == np.int64
    assert np.array_equal(lib.op["tau_sobolev"], np.ascontiguousarray(prob.opacity_state.tau_sobolev).ravel())
    assert lib.cfg["number_of_vpackets"] == n_v and lib.geo["t"] == prob.time_explosion
    vlog = bool(rcfg.ENABLE_VPACKET_TRACKING) and n_v > 0
    _check_against_golden(g, rpc, trackers, hist, vt, eb, el, n_v, vlog)
    assert isinstance(trackers[0].shell_id, int) and isinstance(trackers[0].radius, float)


def _import_reference_solver():
    """modes/classic/solver.py with the two stand-ins its import needs beyond tools/ref_shim.py (the tracking logger, which wants
    the package metadata, and util.base.quantity_linspace, only used by from_config)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_shim

    ref_shim.load_next_rows()  # (HDFWriterMixin / util.base stand-ins, the estimator classes)
    if "tardis.io.logger" not in sys.modules:
        m = types.ModuleType("tardis.io.logger")
        m.montecarlo_tracking = types.SimpleNamespace(log_decorator=lambda f: f)
        sys.modules["tardis.io.logger"] = m
        sys.modules["tardis.io.logger.montecarlo_tracking"] = m.montecarlo_tracking
    ub = sys.modules["tardis.util.base"]
    if not hasattr(ub, "quantity_linspace"):
        ub.quantity_linspace = lambda *a, **k: None
    import tardis.transport.montecarlo.modes.classic.solver as solver

    return solver


@pytest.mark.parametrize("name", ["macroatom_nv3_log", "downbranch_nv2_roulette"])
def test_integration_one_line_rebinding_inside_the_reference_solver(ref, checker_engine, monkeypatch, name):
    """INTEGRATION.md section 1, executed: `solver.montecarlo_transport_with_vpackets = tardis_amd.transport....`, then the
    reference's own MCTransportSolverClassic.run_classic (modes/classic/solver.py:176-273) on the reference's own objects."""
    eng, lib = checker_engine
    solver = _import_reference_solver()
    monkeypatch.setattr(transport, "get_engine", lambda device_id=None: eng)
    monkeypatch.setattr(solver, "montecarlo_transport_with_vpackets", transport.montecarlo_transport_with_vpackets)  # the one line
    prob, g = _golden.load_case(name)
    rpc, rgeo, rop, rcfg = _reference_objects(ref, prob)
    Q = sys.modules["astropy.units"].Quantity
    s = object.__new__(solver.MCTransportSolverClassic)  # (from_config wants a parsed YAML configuration; run_classic reads these)
    s.nthreads = 1
    s.montecarlo_configuration = rcfg
    s.enable_rpacket_tracking = False
    s.spectrum_frequency_grid = Q(prob.spectrum_frequency_grid, "Hz")
    ts = types.SimpleNamespace(packet_collection=rpc, geometry_state_numba=rgeo, opacity_state_numba=rop,
                               time_explosion=Q(prob.time_explosion, "s"))
    hist = s.run_classic(ts, show_progress_bars=False)
    n_v = int(rcfg.NUMBER_OF_VPACKETS)
    vlog = bool(rcfg.ENABLE_VPACKET_TRACKING) and n_v > 0
    assert np.array_equal(hist, g["v_packets_energy_hist"])
    assert np.array_equal(rpc.output_nus, g["output_nus"]) and np.array_equal(rpc.output_energies, g["output_energies"])
    assert np.array_equal(ts.estimators_bulk.mean_intensity_total, g["j_estimator"])
    assert np.array_equal(ts.estimators_line.mean_intensity_blueward[::int(g["line_estimator_stride"])], g["j_blue_estimator"])
    if vlog:
        assert np.array_equal(ts.vpacket_tracker.nus, g["vpacket_nus"]) and np.array_equal(ts.vpacket_tracker.energies, g["vpacket_energies"])
    # the reference's own DataFrame builder ran on the tracker objects the wrapper filled
    # (tracker_last_interaction_util.py:33-134): same table as the engine-side mirror builds from the SoA
    df = ts.tracker_last_interaction_df
    assert len(df) == rpc.number_of_packets
    soa = st.LastInteractionTrackers(rpc.number_of_packets)
    for f in _golden.TRACKER_F64:
        getattr(soa, f)[:] = g["trk_" + f]
    for f in _golden.TRACKER_I64:
        getattr(soa, f)[:] = g["trk_" + f]
    mirror = transport.MonteCarloTransportState(prob.packet_collection, prob.geometry, prob.opacity_state, prob.time_explosion)
    mirror.tracker_last_interaction = soa
    mine = mirror.tracker_last_interaction_df
    assert list(df.columns) == list(mine.columns) and list(df.index.names) == list(mine.index.names)
    for col in df.columns:
        assert df[col].dtype == mine[col].dtype, col
        a, b = df[col].to_numpy(), mine[col].to_numpy()
        if a.dtype.kind == "f":
            assert np.array_equal(a.astype(np.float64), b.astype(np.float64), equal_nan=True), col
        else:
            assert np.array_equal(a, b), col


def test_show_progress_bars_polls_the_engine_and_ends_at_the_packet_count(ref, checker_engine, monkeypatch, capsys):
    """show_progress_bars=True (run_classic's default, modes/classic/solver.py:154-175): the wrapper polls Engine.progress() from a
    thread while the call blocks and finishes the bar at the call's packet count; False polls nothing."""
    eng, lib = checker_engine
    prob, g = _golden.load_case("downbranch_nv0")
    rpc, rgeo, rop, rcfg = _reference_objects(ref, prob)
    seen = []
    real = transport._PacketProgress._update

    def spy(self, final=False):
        real(self, final)
        seen.append((self.seen, final))

    monkeypatch.setattr(transport._PacketProgress, "_update", spy)
    transport.montecarlo_transport_with_vpackets(rpc, rgeo, prob.time_explosion, rop, rcfg, prob.spectrum_frequency_grid, None, 0,
                                                 show_progress_bars=True, packet_propagation_function=ref.packet_propagation, engine=eng)
    assert "progress" in lib.calls and seen and seen[-1] == (rpc.number_of_packets, True)
    assert np.array_equal(rpc.output_nus, g["output_nus"])
    lib.calls.clear(); seen.clear()
    transport.montecarlo_transport_with_vpackets(rpc, rgeo, prob.time_explosion, rop, rcfg, prob.spectrum_frequency_grid, None, 0,
                                                 show_progress_bars=False, packet_propagation_function=ref.packet_propagation, engine=eng)
    assert "progress" not in lib.calls and not seen
