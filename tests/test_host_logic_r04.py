"""CPU tests of the round-4 host logic: residency of the opacity tables is a property of the ENGINE (two solvers on the
process-wide engine must not reuse each other's tables), lazy views of a resident run notice a later reset / all-reduce of
the estimators, and the control plane's rendezvous survives strangers and knows how to meet across nodes."""
import os
import socket
import stat
import struct
import threading
import types

import numpy as np
import pytest

from tardis_amd import distributed, state as st, synthetic, transport


class _FakeEngine:
    """Just enough of tardis_amd.engine.Engine for MCTransportSolverHIP._run_resident on host packets."""

    def __init__(self):
        self.packets_generation = self.results_generation = self.estimators_generation = 0
        self.resident_opacity = None
        self.uploads = []
        self.n_shells, self.n_lines = 3, 5

    def set_geometry(self, geometry, time_explosion=None): pass

    def set_opacity(self, op):
        self.resident_opacity = None
        self.uploads.append(op)
        self.resident_opacity = op

    def set_config(self, *a, **k): pass

    def set_option(self, *a): pass

    def set_packets(self, pc): self.packets_generation += 1

    def reset_estimators(self): self.estimators_generation += 1

    def propagate(self):
        self.results_generation += 2
        self.estimators_generation += 2

    def synchronize(self): pass

    def allreduce_estimators(self): self.estimators_generation += 1

    def last_propagate_ms(self): return 1.0

    def get_results(self, out_nus=None, out_energies=None, **kw):
        n = 4
        return types.SimpleNamespace(j_estimator=np.ones(3), nu_bar_estimator=np.ones(3), counters={}, v_packets_energy_hist=np.zeros(11),
                                     output_nus=out_nus if out_nus is not None else np.ones(n),
                                     output_energies=out_energies if out_energies is not None else np.ones(n),
                                     j_blue_estimator=np.full((5, 3), float(self.estimators_generation)), edotlu_estimator=np.zeros((5, 3)),
                                     trackers=None)

    def radiation_field(self, *a, **k):
        return {"t_radiative": np.ones(3), "dilution_factor": np.ones(3), "j_blues": None}


def _solver(eng):
    grid = np.linspace(1e14, 1e15, 11)
    return transport.MCTransportSolverHIP(grid, resident=True, engine=eng, line_interaction_type="scatter")


def _state(solver, op):
    geo = synthetic.make_geometry(3)
    pc = st.PacketCollection(np.ones(4), np.ones(4), np.ones(4), np.ones(4), np.arange(4), 1.0)
    return solver.initialize_transport_state(pc, geo, op, geo.time_explosion)


def test_opacity_residency_is_tracked_on_the_engine():
    """solver A uploaded opacity X; somebody else -- a second solver, the non-resident entry point -- uploads Y
    into the same engine; solver A's next run on X must upload X again instead of propagating on Y's tables."""
    eng = _FakeEngine()
    a, b = _solver(eng), _solver(eng)
    x, y = object(), object()
    a.run(_state(a, x))
    assert eng.uploads == [x]
    a.run(_state(a, x))
    assert eng.uploads == [x]            # same object, same engine: reused
    b.run(_state(b, y))
    assert eng.uploads == [x, y]
    a.run(_state(a, x))
    assert eng.uploads == [x, y, x]      # round 3 skipped this upload (its cache lived on the solver)
    eng.set_opacity(y)                   # (what montecarlo_transport_with_vpackets does on the shared engine)
    a.run(_state(a, x))
    assert eng.uploads[-1] is x and len(eng.uploads) == 5
    a.reuse_opacity = False
    a.run(_state(a, x))
    assert len(eng.uploads) == 6


def test_lazy_estimator_views_notice_a_reset_or_allreduce():
    """the [L,S] line estimators of a resident run are fetched on first access; a reset_estimators() or an
    all-reduce on the shared engine in between changes what is resident without a new propagate -- the view must say so, and an
    intended all-reduce (N-GPU outer iteration) is declared with estimators_allreduced()."""
    eng = _FakeEngine()
    s = _solver(eng)
    ts = _state(s, object())
    s.run(ts)
    assert ts.radiation_field(np.ones(3))["t_radiative"].shape == (3,)   # host packets: the estimators are still the engine's
    eng.reset_estimators()
    with pytest.raises(RuntimeError, match="reset / all-reduced"):
        ts.estimators_line.mean_intensity_blueward
    with pytest.raises(RuntimeError):
        ts.radiation_field(np.ones(3))
    ts2 = _state(s, object())
    s.run(ts2)
    eng.allreduce_estimators()
    with pytest.raises(RuntimeError):
        ts2.estimators_line.mean_intensity_blueward
    ts2.estimators_allreduced()
    assert ts2.estimators_line.mean_intensity_blueward.shape == (5, 3)
    s.run(_state(s, object()))
    with pytest.raises(RuntimeError, match="propagated again"):
        ts2.radiation_field(np.ones(3))
    with pytest.raises(RuntimeError):
        ts2.estimators_allreduced()      # (a later run cannot be declared this run's)


def test_rendezvous_file_is_private_and_never_follows_links(tmp_path, monkeypatch):
    monkeypatch.setattr(distributed.tempfile, "gettempdir", lambda: str(tmp_path))
    d = distributed._rendezvous_dir()
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    path = distributed._rendezvous_path("29555")
    assert os.path.dirname(path) == d
    distributed._publish_port(path, 4242)
    assert open(path).read() == "4242" and stat.S_IMODE(os.stat(path).st_mode) == 0o600
    distributed._publish_port(path, 4243)   # a stale file of an earlier job (same user, private directory) is replaced
    assert open(path).read() == "4243"
    # a planted symlink where the temporary file goes is removed, not followed
    victim = tmp_path / "victim"
    victim.write_text("untouched")
    os.symlink(victim, f"{path}.{os.getpid()}")
    distributed._publish_port(path, 4244)
    assert victim.read_text() == "untouched" and open(path).read() == "4244"
    # a directory somebody else could write to is refused
    os.chmod(d, 0o777)
    with pytest.raises(PermissionError):
        distributed._rendezvous_dir()
    os.chmod(d, 0o700)


def test_multi_node_launches_take_a_derivable_port(monkeypatch):
    for k in ("LOCAL_WORLD_SIZE", "WORLD_SIZE", "GROUP_RANK", "NODE_RANK", "NNODES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert not distributed._multi_node()
    monkeypatch.setenv("WORLD_SIZE", "16")
    assert distributed._multi_node()          # two nodes of eight: no shared temp dir, no common parent pid
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    monkeypatch.setenv("GROUP_RANK", "1")
    assert distributed._multi_node()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_strangers_on_the_control_port_cost_rank_zero_nothing(monkeypatch):
    """during the rendezvous rank 0 accepts whatever connects.  An HTTP probe, a length prefix of 2**60 bytes,
    a client that says nothing -- each only loses its own connection; the real rank 1 still joins."""
    port = _free_port()
    env = {"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "TARDIS_AMD_CONTROL_PORT": str(port)}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    result = {}

    def rank(r):
        os.environ["RANK"] = str(r)  # (threads of one process: init_from_env reads the environment at entry)
        result[r] = distributed.init_from_env()

    monkeypatch.setenv("RANK", "0")
    t0 = threading.Thread(target=lambda: result.__setitem__(0, distributed.init_from_env()))
    t0.start()
    # strangers first
    def connect():
        for _ in range(200):
            try:
                return socket.create_connection(("127.0.0.1", port), timeout=1.0)
            except OSError:
                import time
                time.sleep(0.02)
        raise AssertionError("rank 0 never listened")
    s1 = connect(); s1.sendall(b"GET / HTTP/1.1\r\nHost: x\r\n\r\n")
    s2 = connect(); s2.sendall(struct.pack("<Q", 1 << 60))
    s3 = connect()  # says nothing (rank 0 times it out on its own; here it is closed right away)
    s3.close()
    # the real peer
    token = distributed._hello_token(int(env["MASTER_PORT"]), 2)
    assert token[:20] == struct.pack("<4sqq", distributed._HELLO, int(env["MASTER_PORT"]), 2) and len(token) == 36
    # a rank of a job with another TARDIS_AMD_CONTROL_TOKEN: right port, right world size, wrong secret
    monkeypatch.setenv("TARDIS_AMD_CONTROL_TOKEN", "somebody else's job")
    other = distributed._hello_token(int(env["MASTER_PORT"]), 2)
    monkeypatch.delenv("TARDIS_AMD_CONTROL_TOKEN")
    assert other != token and other[:20] == token[:20]
    s4 = connect()
    distributed._send_msg(s4, other + struct.pack("<q", 1))
    s4.settimeout(10.0)
    assert s4.recv(1) == b""  # rank 0 closed it without an answer
    s4.close()
    s = connect()
    distributed._send_msg(s, token + struct.pack("<q", 1))
    assert distributed._recv_msg(s, max_bytes=len(token)) == token
    t0.join(timeout=60)
    assert not t0.is_alive() and result[0].world_size == 2 and result[0].rank == 0
    for x in (s1, s2):
        x.close()
    # rank 0's group works with the hand-made peer: one barrier round trip
    th = threading.Thread(target=result[0].barrier)
    th.start()
    distributed._send_msg(s, b"")
    assert distributed._recv_msg(s) == b""
    th.join(timeout=30)
    assert not th.is_alive()
    # after the rendezvous every exchange is bounded by what it can legitimately carry: a peer that lost framing raises
    err = {}

    def reduce_once():
        try:
            result[0].max_float(1.0)
        except ConnectionError as e:
            err["e"] = e
    th = threading.Thread(target=reduce_once)
    th.start()
    s.sendall(struct.pack("<Q", 1 << 40))
    th.join(timeout=30)
    assert not th.is_alive() and "at most 8 bytes" in str(err["e"])
    s.close()
    for p in result[0]._peers:
        p.close()
