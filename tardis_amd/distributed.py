"""Multi-GPU plumbing: one process per GPU, packets sharded by index, ONE all-reduce of the estimator arrays
per Monte Carlo iteration.

The data-path collective on GPUs is RCCL, called inside the engine on its own stream
(tardis_mc_allreduce_estimators).  What is left for the host is a control plane: rendezvous from the launcher's
environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT, as ``python -m torch.distributed.run`` sets
them), broadcast of the 128-byte RCCL unique id, barriers, max-over-ranks of timings, and -- only as the fall-back when
no RCCL communicator exists -- a host-side sum of the estimator arrays.  That is a few small messages per iteration, so
it runs over plain TCP sockets from the standard library (rank 0 is the hub): the package imports neither PyTorch nor
any other framework.  The same code runs on CPU (world_size-2 tests).

Rendezvous on one node.  The launcher's agent already owns MASTER_PORT (its own store), so rank 0 binds an ephemeral
port on MASTER_ADDR and publishes it in a small file named after (MASTER_PORT, the launcher's pid -- every rank is a
child of the same agent) inside a per-user 0700 directory of the temp dir (created O_EXCL | O_NOFOLLOW, removed at exit);
the other ranks poll for it.  ``TARDIS_AMD_CONTROL_PORT`` pins the port instead (ranks started by hand).

Several nodes.  Ranks of other nodes share neither the temp dir nor the parent pid, so a multi-node launch (LOCAL_WORLD_SIZE
!= WORLD_SIZE, or GROUP_RANK / NODE_RANK > 0) takes a fixed port: ``TARDIS_AMD_CONTROL_PORT`` if set, else MASTER_PORT + 1 on
MASTER_ADDR (rank 0 runs on the master node in every torchrun layout).
"""
from __future__ import annotations

import atexit
import hashlib
import os
import socket
import struct
import tempfile
import time
from dataclasses import dataclass, field

import numpy as np

_HELLO = b"TMCG"
_CONNECT_TIMEOUT_S = 300.0
_MAX_MSG_BYTES = 1 << 33  # post-rendezvous messages without a tighter bound of their own (broadcasts): 8 GiB


def _hello_token(master_port: int, world: int) -> bytes:
    """What both ends of a rendezvous connection must present.  (MASTER_PORT, WORLD_SIZE) keeps jobs of one machine apart;
    ``TARDIS_AMD_CONTROL_TOKEN`` (any string, the same on every rank) adds a shared secret for multi-node launches, where
    the port is derivable by anybody who can reach MASTER_ADDR."""
    secret = os.environ.get("TARDIS_AMD_CONTROL_TOKEN", "")
    return struct.pack("<4sqq", _HELLO, int(master_port), world) + hashlib.sha256(b"tardis_amd control plane\0" + secret.encode()).digest()[:16]


def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)))
    if payload:
        sock.sendall(payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("control plane: peer closed the connection")
        got += k
    return bytes(buf)


def _recv_msg(sock: socket.socket, max_bytes: int | None = None) -> bytes:
    """One length-prefixed message.  ``max_bytes`` bounds what an UNTRUSTED peer (anything that connects during the
    rendezvous: a port scanner, an HTTP probe) can make this process allocate."""
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if max_bytes is not None and n > max_bytes:
        raise ConnectionError(f"control plane: a {n}-byte message where at most {max_bytes} bytes are expected")
    return _recv_exact(sock, n) if n else b""


@dataclass
class ProcessGroup:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0
    _peers: list = field(default_factory=list, repr=False)   # rank 0: sockets of ranks 1..W-1 (index r-1)
    _hub: object = field(default=None, repr=False)            # ranks > 0: socket to rank 0
    _rdzv_file: str | None = field(default=None, repr=False)

    @property
    def is_distributed(self) -> bool:
        return self.world_size > 1

    # -- the one primitive: every rank contributes a message, rank 0 reduces, every rank gets the result
    def _allreduce_bytes(self, payload: bytes, reduce, max_bytes: int = _MAX_MSG_BYTES) -> bytes:
        """``max_bytes`` bounds every message of the exchange (contributions and result): a peer that lost framing -- or is
        not the peer it claimed to be -- costs an error, not an allocation of whatever its first eight bytes say."""
        if not self.is_distributed:
            return reduce([payload])
        if self.rank == 0:
            parts = [payload] + [_recv_msg(s, max_bytes) for s in self._peers]
            out = reduce(parts)
            for s in self._peers:
                _send_msg(s, out)
            return out
        _send_msg(self._hub, payload)
        return _recv_msg(self._hub, max_bytes)

    def barrier(self):
        self._allreduce_bytes(b"", lambda parts: b"", max_bytes=0)

    def broadcast_bytes(self, payload: bytes | None, src: int = 0) -> bytes | None:
        """The bytes of rank ``src`` on every rank (None travels as None)."""
        if not self.is_distributed:
            return payload
        mine = (b"\x01" + payload) if (self.rank == src and payload is not None) else (b"\x00" if self.rank == src else b"")
        out = self._allreduce_bytes(mine, lambda parts: parts[src])
        return out[1:] if out[:1] == b"\x01" else None

    def max_float(self, value: float) -> float:
        out = self._allreduce_bytes(struct.pack("<d", float(value)),
                                    lambda parts: struct.pack("<d", max(struct.unpack("<d", p)[0] for p in parts)), max_bytes=8)
        return struct.unpack("<d", out)[0]

    def sum_arrays_(self, arrays):
        """In-place sum over ranks of host float64 arrays, added in rank order (control-plane / CPU path; GPUs use RCCL
        in the engine)."""
        if not self.is_distributed:
            return arrays
        for a in arrays:
            if a.dtype != np.float64 or not a.flags.c_contiguous:
                raise TypeError("sum_arrays_ wants C-contiguous float64 arrays")

            def reduce(parts):
                acc = np.frombuffer(parts[0], dtype=np.float64).copy()
                for p in parts[1:]:
                    acc += np.frombuffer(p, dtype=np.float64)
                return acc.tobytes()

            out = self._allreduce_bytes(a.tobytes(), reduce, max_bytes=a.nbytes)
            a[...] = np.frombuffer(out, dtype=np.float64).reshape(a.shape)
        return arrays

    def destroy(self):
        try:
            if self.is_distributed:
                self.barrier()  # nobody closes while another rank still waits for an answer
        except OSError:
            pass
        for s in self._peers + ([self._hub] if self._hub else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._hub = [], None
        if self._rdzv_file:
            try:
                os.unlink(self._rdzv_file)
            except OSError:
                pass
            self._rdzv_file = None


def _rendezvous_dir() -> str:
    """A directory only this user can write to (the temp dir itself is world-writable: a predictable file name there could
    be pre-created, or be a symlink, by somebody else)."""
    d = os.path.join(tempfile.gettempdir(), f"tardis_amd_ctl_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"control plane: {d} is not a private directory of this user")
    return d


def _rendezvous_path(master_port: str) -> str:
    return os.path.join(_rendezvous_dir(), f"{master_port}_{os.getppid()}")


def _publish_port(path: str, port: int) -> None:
    """Write the port file: never through a symlink, never over somebody else's file; stale files of an earlier job with the
    same (MASTER_PORT, launcher pid) are this user's own (private directory) and are replaced."""
    tmp = f"{path}.{os.getpid()}"
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "w") as f:
        f.write(str(port))
    os.replace(tmp, path)  # (atomic: a reader sees the old content or the new one)


def _unlink_quietly(path: str) -> None:
    try:
        os.unlink(path)
    except OSError:
        pass


def _multi_node() -> bool:
    env = os.environ
    try:
        if int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1"))) != int(env.get("WORLD_SIZE", "1")):
            return True
        return int(env.get("GROUP_RANK", "0")) > 0 or int(env.get("NODE_RANK", "0")) > 0 or int(env.get("NNODES", "1")) > 1
    except ValueError:
        return False


def init_from_env(backend: str = "tcp") -> ProcessGroup:
    """Join the job described by the launcher's environment (``python -m torch.distributed.run`` sets it; so can a shell
    loop).  ``backend`` is accepted for interface stability; the control plane is always the TCP hub described above."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return ProcessGroup(0, 1, local_rank)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    master_port = os.environ.get("MASTER_PORT", "29533")
    fixed = os.environ.get("TARDIS_AMD_CONTROL_PORT")
    if not fixed and _multi_node():
        fixed = str(int(master_port) + 1)  # (no shared temp dir / parent pid across nodes: a port every rank can derive)
    token = _hello_token(int(master_port), world)
    deadline = time.monotonic() + _CONNECT_TIMEOUT_S
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, int(fixed) if fixed else 0))
        srv.listen(world)
        path = None
        if not fixed:
            path = _rendezvous_path(master_port)
            _publish_port(path, srv.getsockname()[1])
            atexit.register(_unlink_quietly, path)  # (destroy() removes it, too; this covers a rank 0 that never gets there)
        peers: list = [None] * (world - 1)
        srv.settimeout(1.0)
        while any(p is None for p in peers):
            if time.monotonic() > deadline:
                raise TimeoutError("control plane: not every rank connected to rank 0")
            try:
                c, _ = srv.accept()
            except socket.timeout:
                continue
            c.settimeout(30.0)
            try:  # (whatever a stray client sends -- garbage, a huge length prefix, nothing -- only costs it its connection)
                hello = _recv_msg(c, max_bytes=len(token) + 8)
                r = struct.unpack("<q", hello[len(token):])[0] if len(hello) == len(token) + 8 and hello[:len(token)] == token else -1
            except Exception:  # noqa: BLE001
                r = -1
            if not (1 <= r < world) or peers[r - 1] is not None:
                c.close()  # (a stray connection, or a rank of another job that read a stale file)
                continue
            c.settimeout(None)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(c, token)
            peers[r - 1] = c
        srv.close()
        return ProcessGroup(0, world, local_rank, peers, None, path)
    while True:
        if time.monotonic() > deadline:
            raise TimeoutError("control plane: rank 0 not reachable")
        s = None
        try:
            port = int(fixed) if fixed else int(open(_rendezvous_path(master_port)).read())
            s = socket.create_connection((addr, port), timeout=5.0)
            s.settimeout(30.0)
            _send_msg(s, token + struct.pack("<q", rank))
            if _recv_msg(s, max_bytes=len(token)) != token:  # (something else listens there: a stale file of an earlier job)
                raise ConnectionError("control plane: wrong peer")
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            return ProcessGroup(rank, world, local_rank, [], s, None)
        except (OSError, ValueError, struct.error):
            if s is not None:  # (every failed attempt gives its socket back)
                try:
                    s.close()
                except OSError:
                    pass
            time.sleep(0.05)


def shard_bounds(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Packet-index range [lo, hi) owned by ``rank`` (contiguous, sizes differ by at most one)."""
    return (rank * n_items) // world_size, ((rank + 1) * n_items) // world_size


def setup_engine_comm(engine, pg: ProcessGroup) -> int:
    """Create the RCCL communicator of ``engine`` across the process group and prove that it spans it.

    Returns the number of ranks the communicator was VERIFIED to sum over -- ``pg.world_size`` on success, on every rank: after
    ``comm_init`` each rank all-reduces a one-element buffer holding rank + 1 and checks N (N + 1) / 2 (``Engine.comm_check``) --
    and 0, on every rank, if the communicator could not be created or failed its check anywhere (librccl missing, two ranks on one
    device, init error); the caller may then reduce the estimators through the control plane (``ProcessGroup.sum_arrays_``) -- and
    has to SAY so: the return value is what a bench line reports as ``rccl_ranks``.  One process: no communicator is needed, and
    none is created; returns 1 (truthy like the world sizes, and what ``rccl_ranks`` means there: one rank, no collective)."""
    if not pg.is_distributed:
        return 1
    uid = None
    if pg.rank == 0:
        try:
            uid = engine.comm_unique_id()
        except Exception as exc:  # noqa: BLE001 -- reported, and agreed on by all ranks below
            print(f"tardis_amd.distributed: no RCCL unique id ({exc})", flush=True)
    uid = pg.broadcast_bytes(uid, src=0)
    failed = 1.0 if uid is None else 0.0
    if uid is not None:
        try:
            engine.comm_init(pg.rank, pg.world_size, uid)
        except Exception as exc:  # noqa: BLE001
            print(f"tardis_amd.distributed: rank {pg.rank}: RCCL communicator not created ({exc})", flush=True)
            failed = 1.0
    if pg.max_float(failed) != 0.0:  # (agreed on BEFORE the check's collective: a rank without a communicator would leave the others hanging in it)
        return 0
    try:
        ok = engine.comm_check() == pg.world_size
    except Exception as exc:  # noqa: BLE001
        print(f"tardis_amd.distributed: rank {pg.rank}: RCCL communicator failed its self-check ({exc})", flush=True)
        ok = False
    return pg.world_size if pg.max_float(0.0 if ok else 1.0) == 0.0 else 0
