"""TEST INFRASTRUCTURE: the ProcessGroup interface of tardis_amd.distributed on torch.distributed's gloo backend.

The product's control plane is the standard-library TCP hub of tardis_amd/distributed.py (the package imports no PyTorch).
This twin exists so that the world_size-2 CPU test also runs the N > 1 host logic (sharding, estimator sum,
communicator set-up agreement) over gloo, the backend the build contract names for CPU multi-process tests."""
import os

import numpy as np


class GlooGroup:
    def __init__(self):
        import torch.distributed as dist

        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        self.dist = dist
        if self.world_size > 1 and not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world_size)

    @property
    def is_distributed(self):
        return self.world_size > 1

    def barrier(self):
        if self.is_distributed:
            self.dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        if not self.is_distributed:
            return payload
        box = [payload]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def max_float(self, value):
        if not self.is_distributed:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_arrays_(self, arrays):
        if not self.is_distributed:
            return arrays
        import torch

        for a in arrays:
            assert isinstance(a, np.ndarray)
            self.dist.all_reduce(torch.from_numpy(a), op=self.dist.ReduceOp.SUM)
        return arrays

    def destroy(self):
        if self.is_distributed and self.dist.is_initialized():
            self.dist.destroy_process_group()
