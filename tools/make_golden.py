"""DEV-CONTAINER-ONLY: generate tests/golden/*.npz by running the REFERENCE itself (pure-Python mode).

    python tools/make_golden.py            # all cases
    python tools/make_golden.py name ...   # selected cases

Each fixture stores the INPUTS the reference ran on (numpy's SIMD transcendental kernels make
tardis_amd.synthetic.make_problem differ in the last bit between CPUs, so inputs are data, not regenerated;
tau_sobolev is stored as its exact rank-1 factors) and the reference's outputs: per-packet output_nus /
output_energies, J, nu_bar, j_blue, Edotlu, the v-packet histogram / log and every TrackerLastInteraction
field.  numpy's AVX-512 transcendental kernels are disabled for the run so that np.log == glibc log, which is
what the Numba-compiled reference calls (see oracle/portable_math.h for the rationale).
"""
import hashlib
import json
import os
import sys
import time

_FEATURES = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
if os.environ.get("NPY_DISABLE_CPU_FEATURES") != _FEATURES:
    os.environ["NPY_DISABLE_CPU_FEATURES"] = _FEATURES
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_shim  # noqa: E402
from tardis_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (make_problem kwargs, extra config)
CASES = {
    "scatter_nv0": (dict(seed=11, n_packets=2000, n_shells=10, n_lines=1500, line_interaction_type="scatter"), {}),
    "downbranch_nv0": (dict(seed=12, n_packets=3000, n_shells=20, n_lines=3000, line_interaction_type="downbranch"), {}),
    "macroatom_nv0": (dict(seed=13, n_packets=2000, n_shells=10, n_lines=2000, line_interaction_type="macroatom"), {}),
    "macroatom_nv3_log": (dict(seed=14, n_packets=500, n_shells=12, n_lines=1200, line_interaction_type="macroatom",
                               n_vpackets=3), dict(ENABLE_VPACKET_TRACKING=True)),
    "downbranch_fullrel": (dict(seed=15, n_packets=1500, n_shells=20, n_lines=2000, line_interaction_type="downbranch",
                                enable_full_relativity=True), {}),
    "macroatom_fullrel_nv2": (dict(seed=16, n_packets=400, n_shells=10, n_lines=1000, line_interaction_type="macroatom",
                                   enable_full_relativity=True, n_vpackets=2), dict(ENABLE_VPACKET_TRACKING=True)),
    "scatter_dense_electrons": (dict(seed=17, n_packets=1500, n_shells=8, n_lines=800, line_interaction_type="scatter",
                                     electron_density_0=2e10), {}),
    "downbranch_nv2_spawnrange": (dict(seed=18, n_packets=500, n_shells=10, n_lines=1000,
                                       line_interaction_type="downbranch", n_vpackets=2,
                                       vpacket_spawn_range=(4.0e14, 1.2e15)), {}),
    "downbranch_thick": (dict(seed=19, n_packets=1000, n_shells=6, n_lines=4000, line_interaction_type="downbranch",
                              log_tau_mean=-1.0), {}),
    # BASELINE.json configs[0] shape (tardis_example: 20 shells, ~3e4 lines, downbranch), fewer packets;
    # line estimators stored for every 16th line + per-shell sums to keep the fixture small
    "config1_downbranch": (dict(seed=1, n_packets=4000, n_shells=20, n_lines=30000, line_interaction_type="downbranch",
                                shell_independent_probabilities=True), dict(_line_estimator_stride=16)),
    "scatter_single_shell": (dict(seed=20, n_packets=800, n_shells=1, n_lines=300, line_interaction_type="scatter"), {}),
    # Russian roulette with survivors (virtual_packet.py:221-232): optically thick lines so that tau > VPACKET_TAU_RUSSIAN occurs
    "downbranch_nv2_roulette": (dict(seed=22, n_packets=400, n_shells=8, n_lines=1500, line_interaction_type="downbranch",
                                     n_vpackets=2, log_tau_mean=-0.5), dict(SURVIVAL_PROBABILITY=0.3, ENABLE_VPACKET_TRACKING=True)),
    # heavy-tailed macro-atom blocks (macroatom_solver.py:383-428,624-670: a block is ALL transitions out of one level):
    # blocks of 33 ... 2100 rows, probabilities spread over many decades (most rows below 2**-16 of the block sum)
    "macroatom_heavy_nv0": (dict(seed=31, n_packets=1200, n_shells=10, n_lines=4000, line_interaction_type="macroatom",
                                 level_sizes="heavy", log_tau_mean=-2.0), {}),
    "downbranch_heavy_nv2": (dict(seed=32, n_packets=800, n_shells=10, n_lines=4000, line_interaction_type="downbranch",
                                  n_vpackets=2, level_sizes="heavy", log_tau_mean=-2.0), dict(ENABLE_VPACKET_TRACKING=True)),
    "macroatom_heavy_fullrel_nv2": (dict(seed=33, n_packets=400, n_shells=8, n_lines=4000, line_interaction_type="macroatom",
                                         n_vpackets=2, level_sizes="heavy", log_tau_mean=-2.0, enable_full_relativity=True),
                                    dict(ENABLE_VPACKET_TRACKING=True)),
    # the BASELINE configs[4] combination at fixture size: 100 shells, macroatom on heavy-tailed blocks, ten v-packets per volley
    # (each crossing up to 100 shells), consolidated v-packet log
    "macroatom_heavy_100shells_nv10": (dict(seed=34, n_packets=400, n_shells=100, n_lines=3000, line_interaction_type="macroatom",
                                            n_vpackets=10, level_sizes="heavy", log_tau_mean=-2.5, shell_independent_probabilities=True),
                                       dict(ENABLE_VPACKET_TRACKING=True)),
    # quirk (iii) of SURVEY 8a: disable_line_scattering with non-zero tau_sobolev (real runs zero tau first, opacity_solver.py:46-56)
    "scatter_disabled_lines_tau": (dict(seed=21, n_packets=300, n_shells=6, n_lines=400, line_interaction_type="scatter",
                                        disable_line_scattering=True), {}),
}


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def input_digests(prob) -> dict:
    pc, op, geo = prob.packet_collection, prob.opacity_state, prob.geometry
    return {
        "packets": digest(pc.initial_radii, pc.initial_nus, pc.initial_mus, pc.initial_energies, pc.packet_seeds),
        "geometry": digest(geo.r_inner, geo.r_outer, np.float64(geo.time_explosion)),
        "opacity": digest(op.electron_density, op.line_list_nu, op.tau_sobolev, op.transition_probabilities,
                          op.line2macro_level_upper, op.macro_block_edge_index, op.transition_type,
                          op.destination_level_id, op.transition_line_id),
        "grid": digest(prob.spectrum_frequency_grid),
    }


def to_reference_objects(ref, prob):
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


def run_case(ref, name):
    kwargs, extra = CASES[name]
    extra = dict(extra)
    prob = synthetic.make_problem(**kwargs)
    stride = extra.pop("_line_estimator_stride", 1) if "_line_estimator_stride" in extra else 1
    extra = {k: v for k, v in extra.items() if not k.startswith("_")}
    for k, v in extra.items():
        setattr(prob.montecarlo_configuration, k, v)
    rpc, rgeo, rop, rcfg = to_reference_objects(ref, prob)
    P = rpc.number_of_packets
    trackers = [ref.TrackerLastInteraction() for _ in range(P)]
    t0 = time.time()
    hist, vtracker, est_bulk, est_line = ref.montecarlo_transport_with_vpackets(
        rpc, rgeo, prob.time_explosion, rop, rcfg, prob.spectrum_frequency_grid, trackers,
        rcfg.NUMBER_OF_VPACKETS, False, ref.packet_propagation)
    dt = time.time() - t0
    pc, op, geo, cfg = prob.packet_collection, prob.opacity_state, prob.geometry, prob.montecarlo_configuration
    tau0, rho = op.tau_factors
    if not np.array_equal(tau0[:, None] * rho[None, :], op.tau_sobolev):  # store in full when not exactly rank-1
        tau0, rho = op.tau_sobolev, np.zeros(0)
    out = dict(
        make_problem_kwargs=json.dumps(kwargs), config_extra=json.dumps(extra),
        config=json.dumps({k: (v if not isinstance(v, np.ndarray) else None) for k, v in vars(cfg).items()}),
        in_initial_radii=pc.initial_radii, in_initial_nus=pc.initial_nus, in_initial_mus=pc.initial_mus,
        in_initial_energies=pc.initial_energies, in_packet_seeds=pc.packet_seeds,
        in_radiation_field_luminosity=pc.radiation_field_luminosity,
        in_r_inner=geo.r_inner, in_r_outer=geo.r_outer, in_v_inner=geo.v_inner, in_v_outer=geo.v_outer,
        in_time_explosion=geo.time_explosion,
        in_electron_density=op.electron_density, in_line_list_nu=op.line_list_nu, in_tau0=tau0, in_tau_rho=rho,
        in_transition_probabilities=(op.transition_probabilities[:, :1]
                                     if kwargs.get("shell_independent_probabilities") else op.transition_probabilities),
        in_line2macro_level_upper=op.line2macro_level_upper, in_macro_block_edge_index=op.macro_block_edge_index,
        in_transition_type=op.transition_type, in_destination_level_id=op.destination_level_id,
        in_transition_line_id=op.transition_line_id, in_spectrum_frequency_grid=prob.spectrum_frequency_grid,
        output_nus=rpc.output_nus, output_energies=rpc.output_energies,
        j_estimator=est_bulk.mean_intensity_total, nu_bar_estimator=est_bulk.mean_frequency,
        j_blue_estimator=est_line.mean_intensity_blueward[::stride],
        edotlu_estimator=est_line.energy_deposition_line_rate[::stride],
        line_estimator_stride=stride,
        j_blue_shell_sums=est_line.mean_intensity_blueward.sum(axis=0),
        edotlu_shell_sums=est_line.energy_deposition_line_rate.sum(axis=0),
        v_packets_energy_hist=hist,
    )
    f64 = ("radius", "nu", "mu", "energy", "before_nu", "before_mu", "before_energy", "after_nu", "after_mu",
           "after_energy")
    i64 = ("shell_id", "interaction_type", "interaction_line_absorb_id", "interaction_line_emit_id",
           "interactions_count")
    for f in f64:
        out["trk_" + f] = np.array([getattr(t, f) for t in trackers], dtype=np.float64)
    for f in i64:
        out["trk_" + f] = np.array([getattr(t, f) for t in trackers], dtype=np.int64)
    if rcfg.ENABLE_VPACKET_TRACKING and rcfg.NUMBER_OF_VPACKETS > 0:
        out.update(vpacket_nus=vtracker.nus, vpacket_energies=vtracker.energies,
                   vpacket_initial_mus=vtracker.initial_mus, vpacket_initial_rs=vtracker.initial_rs)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    emitted = int(np.sum(rpc.output_energies >= 0))
    print(f"{name}: P={P} {dt:.1f}s emitted={emitted} size={os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def leaf_kats(ref):
    """Known answers of the reference's leaf functions on the inputs of its own unit tests
    (transport/montecarlo/tests/conftest.py:141-284, tests/test_transport.py:29-306, packets/tests/test_packet.py)."""
    import tardis.transport.montecarlo.modes.homologous_rad_packet_transport as rt
    from tardis.transport.montecarlo.estimators.estimators_bulk import init_estimators_bulk
    from tardis.transport.montecarlo.estimators.estimators_line import init_estimators_line
    from tardis.transport.montecarlo.interaction_event_callers import line_scatter_event
    from tardis.transport.montecarlo.interaction_events import thomson_scatter
    from tardis.transport.montecarlo.packets.movement import move_r_packet
    from tardis.transport.montecarlo.packets.packet_collections import VPacketCollection
    from tardis.transport.montecarlo.packets.radiative_packet import RPacket
    from tardis.transport.montecarlo.packets.virtual_packet import trace_vpacket_volley

    kats = []
    t_exp = 5.2e7

    def opacity(nus, taus, n_e, macro=None):
        L, S = np.asarray(taus).shape
        if macro is None:
            macro = (np.zeros((1, S)), np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64),
                     np.zeros(1, np.int64), np.zeros(1, np.int64))
        return dict(electron_density=list(n_e), line_list_nu=list(nus), tau_sobolev=np.asarray(taus).tolist(),
                    transition_probabilities=np.asarray(macro[0]).tolist(), line2macro_level_upper=list(map(int, macro[1])),
                    macro_block_edge_index=list(map(int, macro[2])), transition_type=list(map(int, macro[3])),
                    destination_level_id=list(map(int, macro[4])), transition_line_id=list(map(int, macro[5])))

    def ref_opacity(o):
        return ref.OpacityStateNumba(
            np.array(o["electron_density"]), np.zeros(len(o["electron_density"])), np.array(o["line_list_nu"]),
            np.array(o["tau_sobolev"]), np.array(o["transition_probabilities"]),
            np.array(o["line2macro_level_upper"], np.int64), np.array(o["macro_block_edge_index"], np.int64),
            np.array(o["transition_type"], np.int64), np.array(o["destination_level_id"], np.int64),
            np.array(o["transition_line_id"], np.int64), np.zeros(0), np.zeros((0, 0)), np.zeros(0), np.zeros(0),
            np.zeros(0, np.int64), np.zeros((0, 0)), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros((0, 0)),
            np.zeros(0, np.int64), -1)

    def ref_geometry(g):
        r_i, r_o = np.array(g["r_inner"]), np.array(g["r_outer"])
        return ref.NumbaHomologousRadial1DGeometry(r_i, r_o, r_i / g["time_explosion"], r_o / g["time_explosion"],
                                                   g["time_explosion"])

    def mk_packet(pk):
        p = RPacket(pk["r"], pk["mu"], pk["nu"], pk["energy"], pk.get("seed", 0), 0)
        p.next_line_id = pk.get("next_line_id", 0)
        p.current_shell_id = pk.get("shell", 0)
        return p

    def pk_out(p):
        return dict(r=p.r, mu=p.mu, nu=p.nu, energy=p.energy, next_line_id=int(p.next_line_id),
                    shell=int(p.current_shell_id), status=int(p.status))

    base_pkt = dict(r=7.5e14, mu=0.3, nu=4e14, energy=0.9)
    # --- trace_packet
    trace_cases = [
        ("boundary", [3.95e14, 3.90e14], [[0.0], [0.0]], [7e14], [8e14], 1e-20 / 6.652458734e-25, False, False),
        ("escatter", [3.95e14, 3.90e14], [[0.0], [0.0]], [7e14], [8e14], 1e-12 / 6.652458734e-25, False, False),
        ("line", [3.999e14, 3.998e14], [[100.0], [100.0]], [7e14], [2e16], 1e-20 / 6.652458734e-25, False, False),
        ("line_disabled", [3.999e14, 3.998e14], [[100.0], [100.0]], [7e14], [2e16], 1e-20 / 6.652458734e-25, False, True),
        ("line_fullrel", [3.999e14, 3.998e14], [[100.0], [100.0]], [7e14], [2e16], 1e-20 / 6.652458734e-25, True, False),
        ("close_line", [4e14 * (1 - 0.3 * 7.5e14 / t_exp / 2.99792458e10), 3.9e14], [[0.5], [0.2]], [7e14], [8e14],
         1e-20 / 6.652458734e-25, False, False),
    ]
    for name, nus, taus, r_in, r_out, n_e, full, dls in trace_cases:
        g = dict(r_inner=r_in, r_outer=r_out, time_explosion=t_exp)
        o = opacity(nus, taus, [n_e])
        chi = n_e * 6.652458734e-25
        p = mk_packet(base_pkt)
        el = init_estimators_line((len(nus), 1))
        np.random.seed(1963)
        d, it, ds = rt.trace_packet(p, ref_geometry(g), t_exp, ref_opacity(o), el, chi, 1.0, False, full, dls)
        kats.append(dict(kind="trace_packet", name=name, packet=base_pkt, seed=1963, chi=chi, geometry=g, opacity=o,
                         full_relativity=full, disable_line_scattering=dls,
                         expect=dict(distance=d, interaction_type=int(it), delta_shell=int(ds), packet=pk_out(p),
                                     j_blue=el.mean_intensity_blueward.tolist(),
                                     edotlu=el.energy_deposition_line_rate.tolist())))
    # --- move_r_packet
    g2 = dict(r_inner=[6.912e14, 8.64e14], r_outer=[8.64e14, 1.0368e15], time_explosion=t_exp)
    for full in (False, True):
        pk = dict(base_pkt, shell=1)
        p = mk_packet(pk)
        eb = init_estimators_bulk(2)
        move_r_packet(p, 1.0e13, ref_geometry(g2), eb, full)
        kats.append(dict(kind="move_r_packet", name=f"full{int(full)}", packet=pk, distance=1.0e13, geometry=g2,
                         full_relativity=full,
                         expect=dict(packet=pk_out(p), j=eb.mean_intensity_total.tolist(), nu_bar=eb.mean_frequency.tolist())))
    # --- thomson_scatter
    for full in (False, True):
        p = mk_packet(base_pkt)
        np.random.seed(1963)
        thomson_scatter(p, t_exp, full)
        kats.append(dict(kind="thomson_scatter", name=f"full{int(full)}", packet=base_pkt, seed=1963, geometry=g2,
                         full_relativity=full, expect=dict(packet=pk_out(p))))
    # --- line_scatter_event with a small macro atom (2 levels, 3 lines)
    nus = [4.2e14, 4.0e14, 3.8e14]
    taus = [[1.0, 2.0], [0.5, 0.7], [0.1, 0.2]]
    macro = (np.array([[0.2, 0.1], [0.3, 0.3], [0.5, 0.6], [0.6, 0.5], [0.4, 0.5]]),  # probs [T=5, S=2]
             [0, 1, 1], [0, 3, 5], [-1, 1, 0, -1, -1], [-99, 1, 0, -99, -99], [0, 1, 2, 1, 2])
    o3 = opacity(nus, taus, [1e9, 5e8], macro)
    for lit in (0, 1, 2):
        for full in (False, True):
            for seed in (1963, 1, 2111963, 10000):
                pk = dict(base_pkt, next_line_id=1, shell=1)
                p = mk_packet(pk)
                np.random.seed(seed)
                line_scatter_event(p, t_exp, lit, ref_opacity(o3), full)
                kats.append(dict(kind="line_scatter_event", name=f"lit{lit}_full{int(full)}_seed{seed}", packet=pk,
                                 seed=seed, geometry=g2, opacity=o3, line_interaction_type=lit, full_relativity=full,
                                 expect=dict(packet=pk_out(p))))
    # --- v-packet volley
    grid = np.linspace(1e14, 1e15, 11)
    for full in (False, True):
        for r0, shell in ((6.912e14, 0), (9.0e14, 1)):
            pk = dict(base_pkt, r=r0, next_line_id=2, shell=shell)
            p = mk_packet(pk)
            vc = VPacketCollection(0, grid, 0.0, 1e200, 4, 4)
            np.random.seed(23)
            trace_vpacket_volley(p, vc, ref_geometry(g2), t_exp, ref_opacity(o3), full, 10.0, 0.0)
            kats.append(dict(kind="trace_vpacket_volley", name=f"full{int(full)}_shell{shell}", packet=pk, seed=23,
                             geometry=g2, opacity=o3, n_vpackets=4, full_relativity=full,
                             expect=dict(nus=vc.nus[:vc.idx].tolist(), energies=vc.energies[:vc.idx].tolist(),
                                         mus=vc.initial_mus[:vc.idx].tolist())))
    path = os.path.join(OUT, "leaf_kats.json")
    with open(path, "w") as f:
        json.dump(kats, f, indent=1)
    print(f"leaf_kats.json: {len(kats)} cases")

    # libm log/exp probes of the generating machine (lets tests decide whether bit-exactness vs libm is expected)
    import math
    rng = np.random.default_rng(99)
    x = rng.random(2000)
    np.savez_compressed(os.path.join(OUT, "libm_probe.npz"), x=x, log_x=np.array([math.log(v) for v in x]),
                        exp_mx=np.array([math.exp(-30 * v) for v in x]))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shim.load()
    names = [a for a in sys.argv[1:]] or (list(CASES) + ["leaf"])
    for n in names:
        if n == "leaf":
            leaf_kats(ref)
        else:
            run_case(ref, n)


if __name__ == "__main__":
    main()
