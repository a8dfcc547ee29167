"""Data parallelism on the GPU path: two replicas against one process.  On a box with at least two GPUs the replicas are ONE RANK PER
DEVICE over RCCL (backend "nccl": the product's real data-parallel path, so that a multi-GPU driver box runs RCCL through these parity
tests before bench.py does -- VERDICT r05 item 5a); on a one-GPU box the two processes share the GPU over a gloo rendezvous.

* dp_mode="global" (all-gather of the fused features, normalisation + losses on the whole batch,
  gradients summed): 2 replicas x 4 clips must reproduce 1 device x 8 clips -- same loss, same gradient.
* dp_mode="replica" (the reference's MirroredStrategy, mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:342-349): the applied
  gradient is the mean of the two per-slice gradients.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def dist_backend(world=2):
    """'nccl' (RCCL, one rank per GPU) when the box has a GPU per rank, else 'gloo' (the ranks share the box's GPU)."""
    return "nccl" if torch.cuda.device_count() >= world else "gloo"


def _run_replicas(mode, tmp_path, world=2, overlap="0"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / ("dp_%s_%s.npz" % (mode, overlap)))
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r), UGN_DP_BACKEND=dist_backend(world), UGN_AR_OVERLAP=overlap)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dp_worker.py"), mode, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    return np.load(out)


def _single(rows):
    import dp_worker as W
    xs, uses, labels, onehot = W.job_batch()
    core = W.make_core(1, "replica")
    cut = lambda a: torch.from_numpy(np.ascontiguousarray(a[rows])).cuda()
    core.forward_backward([cut(x) for x in xs], [cut(u) for u in uses], labels[rows], cut(onehot))
    return core.store.grad.cpu().numpy().copy(), core.losses()


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.timeout(900)
def test_global_batch_replicas_equal_one_device(dev, tmp_path):
    got = _run_replicas("global", tmp_path)
    grad, losses = _single(slice(0, 8))
    assert abs(float(got["loss"]) - losses["loss"]) <= 1e-6 * max(1.0, abs(losses["loss"]))
    assert abs(float(got["triplet"]) - losses["triplet"]) <= 1e-6
    assert _rel(got["grad"], grad) <= 1e-5      # only the order of the fp32 sums over clips differs
    assert np.isfinite(got["params"]).all()


@pytest.mark.timeout(900)
def test_replica_mode_averages_the_slice_gradients(dev, tmp_path):
    got = _run_replicas("replica", tmp_path)
    g0, l0 = _single(slice(0, 4))
    g1, l1 = _single(slice(4, 8))
    assert _rel(got["grad"], 0.5 * (g0 + g1)) <= 1e-6
    assert abs(float(got["loss"]) - l0["loss"]) <= 1e-6 * max(1.0, abs(l0["loss"]))   # rank 0 reports its own slice


@pytest.mark.timeout(900)
def test_bucketed_allreduce_equals_the_single_one(dev, tmp_path):
    """The gradient leaves in buckets (head, then one per branch) while the backward pass still runs; the reduced gradient and
    the updated parameters are those of one all-reduce after the backward pass, bit for bit."""
    a = _run_replicas("replica", tmp_path, overlap="1")
    b = _run_replicas("replica", tmp_path, overlap="0")
    assert np.array_equal(a["grad"], b["grad"]) and np.array_equal(a["params"], b["params"])


@pytest.mark.timeout(900)
def test_bench_contract_with_two_ranks(dev):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one JSON line from rank 0): one rank per GPU over RCCL where
    the box has two GPUs, else rehearsed over gloo with both ranks on the box's one GPU (UGN_DIST_BACKEND=gloo)."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UGN_DIST_BACKEND=dist_backend(2))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--dense-only"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "dp2" and d["roofline"]["achieved"] > 0
    assert d["config"]["distributed"]["backend"] == dist_backend(2) and d["config"]["distributed"]["world_size"] == 2
