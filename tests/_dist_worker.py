"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 job (launched by torch.distributed.run or by hand);
argv[2] picks the control plane: "tcp" = the product's own (tardis_amd.distributed, standard library only), "gloo" = the
torch.distributed twin of tests/_gloo_group.py.

Exercises the host-side multi-GPU plumbing on CPU: environment rendezvous, packet sharding by index, disjoint
per-packet output slices and the sum-all-reduce of the estimator arrays.  The per-shard transport itself is run by
the CPU oracle here (test infrastructure) because no GPU exists in this container."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from tardis_amd import distributed, synthetic  # noqa: E402


def main():
    out_path = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "tcp"
    if backend == "gloo":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _gloo_group import GlooGroup
        pg = GlooGroup()
    else:
        pg = distributed.init_from_env()
        assert "torch" not in sys.modules  # the product's control plane is standard library only
    assert pg.world_size == 2
    prob = synthetic.make_problem(seed=31, n_packets=3001, n_shells=6, n_lines=900, line_interaction_type="macroatom")
    pc = prob.packet_collection
    lo, hi = distributed.shard_bounds(pc.number_of_packets, pg.rank, pg.world_size)
    shard = pc.shard(pg.rank, pg.world_size)
    assert shard.number_of_packets == hi - lo
    r = oracle.run(shard, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
                   prob.spectrum_frequency_grid, math_mode=oracle.MATH_PORTABLE, write_outputs_in_place=True)
    est = [r.j_estimator, r.nu_bar_estimator, r.j_blue_estimator, r.edotlu_estimator, r.v_packets_energy_hist]
    pg.sum_arrays_(est)
    # control-plane helpers
    payload = pg.broadcast_bytes(b"x" * 128 if pg.rank == 0 else None, src=0)
    assert payload == b"x" * 128
    assert pg.max_float(float(pg.rank)) == 1.0
    assert pg.broadcast_bytes(b"from one" if pg.rank == 1 else None, src=1) == b"from one"
    assert pg.broadcast_bytes(None, src=0) is None

    # communicator set-up: every rank learns the same verdict, whichever rank the failure happens on.  The fake engine's collective runs
    # over the control plane (a checker standing where RCCL stands): comm_check is Engine.comm_check's contract -- the sum of rank + 1
    # over the communicator must be N (N + 1) / 2
    class FakeEngine:
        def __init__(self, fail_uid=False, fail_init_on=None, short_on=None, check_raises_on=None):
            self.fail_uid, self.fail_init_on, self.short_on, self.check_raises_on, self.inited = fail_uid, fail_init_on, short_on, check_raises_on, None

        def comm_unique_id(self):
            if self.fail_uid:
                raise RuntimeError("no librccl")
            return b"u" * 128

        def comm_init(self, rank, world_size, uid):
            if self.fail_init_on == rank:
                raise RuntimeError("init failed")
            self.inited = (rank, world_size, uid)

        def comm_check(self):
            rank, world, _ = self.inited
            cell = np.array([float(rank + 1)])
            pg.sum_arrays_([cell])  # (every rank takes part in the collective, as with RCCL)
            if self.check_raises_on == rank:
                raise RuntimeError("ncclAllReduce failed")
            if self.short_on == rank:  # a communicator that reached fewer ranks than the job has (what a silent fall-back would hide)
                cell[0] = float(rank + 1)
            if cell[0] != 0.5 * world * (world + 1):
                raise RuntimeError(f"communicator self-check: sum {cell[0]}")
            return world

    good = FakeEngine()
    assert distributed.setup_engine_comm(good, pg) == 2 and good.inited == (pg.rank, 2, b"u" * 128)
    assert distributed.setup_engine_comm(FakeEngine(fail_uid=True), pg) == 0
    assert distributed.setup_engine_comm(FakeEngine(fail_init_on=1), pg) == 0
    assert distributed.setup_engine_comm(FakeEngine(short_on=1), pg) == 0
    assert distributed.setup_engine_comm(FakeEngine(check_raises_on=0), pg) == 0
    pg.barrier()
    np.savez(out_path + f".rank{pg.rank}.npz", lo=lo, hi=hi, output_nus=pc.output_nus[lo:hi],
             output_energies=pc.output_energies[lo:hi], j=est[0], nu_bar=est[1], j_blue=est[2], edotlu=est[3])
    pg.destroy()


if __name__ == "__main__":
    main()
