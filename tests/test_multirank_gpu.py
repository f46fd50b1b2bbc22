"""The N > 1 path of bench.py on real hardware, as far as one GPU allows: two ranks (one process each, launched the way the
driver launches them) share GPU 0.  RCCL refuses two ranks on one device, so the job takes bench.py's documented fall-back --
the estimator arrays are summed through the control plane (the package's TCP hub, tardis_amd/distributed.py) -- which exercises everything else of the N > 1 path: the
rendezvous, rank r drawing packets [r P, (r+1) P) of the 2 P-packet stream on the device, barriers, max-over-ranks timing.
The summed estimators must equal those of ONE rank propagating all 2 P packets (results are partition-invariant up
to the summation order)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_match_a_single_rank(tmp_path):
    P = 100_000
    common = ["--config", "2", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--boundary-packets", "0"]
    two = tmp_path / "two.npz"
    one = tmp_path / "one.npz"
    line2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--packets", str(P), "--all-on-device", "0",
                  "--dump-estimators", str(two)] + common)
    line1 = _run([sys.executable, "bench.py", "--gpus", "1", "--packets", str(2 * P), "--dump-estimators", str(one)] + common)
    assert line2["n_gpus"] == 2 and line2["scaling"] == "weak" and line2["config"]["packets_per_gpu"] == P
    assert line2["value"] > 0 and line1["n_gpus"] == 1
    # two ranks on ONE device: RCCL refuses, every rank agrees on that, the line says so in both places (rccl_ranks = ranks the all-reduce was
    # verified to span by the (rank + 1) self-check, 0 = host fall-back); the single-rank line carries the field, too
    assert line2["rccl_ranks"] == 0 and "host-side sum of estimators" in line2["config"]["parallelism"]
    assert line1["rccl_ranks"] == 1
    a, b = np.load(two), np.load(one)
    for k in ("j_estimator", "nu_bar_estimator", "j_blue_shell_sums", "edotlu_shell_sums", "j_blue_line_sums"):
        assert_allclose(a[k], b[k], rtol=1e-10, err_msg=k)
    assert np.all(a["j_estimator"] > 0)
    # --scaling strong: the packet count is the job's (BASELINE configs[3]: 1e8 packets shared by 8 GPUs), every rank takes its share
    strong = tmp_path / "strong.npz"
    line3 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--packets", str(2 * P), "--scaling", "strong",
                  "--all-on-device", "0", "--dump-estimators", str(strong)] + common)
    assert line3["scaling"] == "strong" and line3["config"]["packets_per_gpu"] == P and "strong scaling" in line3["config"]["parallelism"]
    c = np.load(strong)
    for k in ("j_estimator", "nu_bar_estimator", "j_blue_shell_sums"):
        assert_allclose(c[k], b[k], rtol=1e-10, err_msg=k)


def test_eight_ranks_launch_path_on_one_gpu(tmp_path):
    """The N = 8 launch of the driver's scaling bench, run once for real: eight processes started by torch.distributed.run, the TCP
    control plane with eight ranks, rank r drawing packets [r P, (r+1) P) of the 8 P-packet stream, the agreement that RCCL is not
    available (eight ranks on one device), the host-side estimator sum, max-over-ranks timing, ONE JSON line from rank 0 -- and
    the summed estimators of the eight shards equal one rank's run over all 8 P packets."""
    P = 20_000
    common = ["--config", "2", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--boundary-packets", "0"]
    eight = tmp_path / "eight.npz"
    one = tmp_path / "one.npz"
    line8 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--packets", str(P), "--all-on-device", "0",
                  "--dump-estimators", str(eight)] + common, timeout=1500)
    line1 = _run([sys.executable, "bench.py", "--gpus", "1", "--packets", str(8 * P), "--dump-estimators", str(one)] + common)
    assert line8["n_gpus"] == 8 and line8["config"]["packets_per_gpu"] == P and line8["value"] > 0
    assert "x8" in line8["config"]["parallelism"]
    assert line8["rccl_ranks"] == 0 and "host-side sum of estimators" in line8["config"]["parallelism"]
    a, b = np.load(eight), np.load(one)
    for k in ("j_estimator", "nu_bar_estimator", "j_blue_shell_sums", "edotlu_shell_sums", "j_blue_line_sums"):
        assert_allclose(a[k], b[k], rtol=1e-10, err_msg=k)
    assert line1["n_gpus"] == 1


def test_require_rccl_turns_the_fall_back_into_a_failure():
    """--require-rccl 1 (the default whenever every rank has a GPU of its own, i.e. without --all-on-device): a job whose communicator does not span its
    ranks prints NO line and exits with rc 3 -- on an 8-GPU node a silent host fall-back cannot pass for an RCCL measurement."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--packets", "20000", "--all-on-device", "0", "--require-rccl", "1",
                        "--config", "2", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--boundary-packets", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "--require-rccl" in r.stderr and "rccl_ranks = 0" in r.stderr
