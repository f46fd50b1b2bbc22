"""Multi-GPU plumbing: one process per GPU, packets sharded by index, ONE all-reduce of the estimator arrays
per Monte Carlo iteration.

The data-path collective on GPUs is RCCL, called inside the engine on its own stream
(tardis_mc_allreduce_estimators).  ``torch.distributed`` is used only as the control plane: rendezvous from the
launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT), broadcast of the 128-byte
RCCL unique id, barriers, and max-over-ranks of timings.  The same helpers run on CPU with the gloo backend
(world_size-2 tests).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class ProcessGroup:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0
    dist: object = None  # torch.distributed module when world_size > 1

    @property
    def is_distributed(self) -> bool:
        return self.world_size > 1

    def barrier(self):
        if self.is_distributed:
            self.dist.barrier()

    def broadcast_bytes(self, payload: bytes | None, src: int = 0) -> bytes:
        if not self.is_distributed:
            return payload
        box = [payload]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def max_float(self, value: float) -> float:
        if not self.is_distributed:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_arrays_(self, arrays):
        """In-place sum over ranks of host numpy arrays (control-plane / CPU path; GPUs use RCCL in the engine)."""
        if not self.is_distributed:
            return arrays
        import torch

        for a in arrays:
            t = torch.from_numpy(a)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return arrays

    def destroy(self):
        if self.is_distributed and self.dist.is_initialized():
            self.dist.destroy_process_group()


def init_from_env(backend: str = "gloo") -> ProcessGroup:
    """Join the job described by the launcher's environment (torch.distributed.run sets it)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return ProcessGroup(0, 1, local_rank, None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch.distributed as dist

    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return ProcessGroup(rank, world, local_rank, dist)


def shard_bounds(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Packet-index range [lo, hi) owned by ``rank`` (contiguous, sizes differ by at most one)."""
    return (rank * n_items) // world_size, ((rank + 1) * n_items) // world_size


def setup_engine_comm(engine, pg: ProcessGroup) -> bool:
    """Create the RCCL communicator of ``engine`` across the process group (no-op for one process).

    Returns False -- on every rank -- if the communicator could not be created anywhere (librccl missing, init error); the
    caller can then reduce the estimators through the control plane (``ProcessGroup.sum_arrays_``) instead of failing."""
    if not pg.is_distributed:
        return True
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
    return pg.max_float(failed) == 0.0
