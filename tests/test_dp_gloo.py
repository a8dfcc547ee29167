"""The N > 1 path on CPU: two processes over gloo exercise the same sharding + gradient all-reduce helpers the GPU
engine uses with RCCL, with the CPU oracle standing in for the per-replica compute.  Expected semantics = the
reference's MirroredStrategy: per-replica loss on the local slice, gradients averaged over replicas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ugaitnet_oracle as O
from tests.synth import make_batch
from ugaitnet_amd import dp

KINDS, B, L, NCLS = ("of", "gray"), 8, 2, 4


def _params():
    rng = np.random.default_rng(9)
    return dict(branches=[O.init_branch_params(rng, 2 if k == 'of' else 1, np.float64) for k in KINDS],
                head=O.init_head_params(rng, NCLS, np.float64))


def _flat(g):
    parts = [g['branches'][i][k].ravel() for i in range(len(KINDS)) for k in sorted(g['branches'][i])]
    parts += [g['head'][k].ravel() for k in sorted(g['head'])]
    return np.concatenate(parts)


def _replica_grads(rank, world):
    xs, uses, labels, onehot = make_batch(KINDS, B, L, NCLS, ids=4, seed=31, dtype=np.float64)
    X = [xs[0], uses[0], xs[1], uses[1]]
    Xs, ys = dp.shard_batch(X, [labels.reshape(-1, 1), onehot], rank, world)
    _, g = O.model_loss_and_grads([Xs[0], Xs[2]], [Xs[1], Xs[3]], ys[0].reshape(-1), ys[1], _params())
    return _flat(g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    flat = torch.from_numpy(_replica_grads(rank, world))
    scale = dp.allreduce_sum_(flat)
    out[rank] = (flat * scale).numpy()
    # global-batch mode's exchange: equal slices concatenated along the batch axis in rank order; ragged slices are refused
    part = torch.full((3, 2, 4), float(rank)) + torch.arange(4.0)
    whole = dp.gather_batch_axis(part, 1)
    out["gather%d" % rank] = whole.numpy()
    try:
        dp.gather_batch_axis(torch.zeros(3, 2 + rank, 4), 1)
        out["ragged%d" % rank] = "accepted"
    except ValueError:
        out["ragged%d" % rank] = "refused"
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    for n in (24, 40, 7):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


@pytest.mark.timeout(300)
def test_two_replicas_average_gradients_over_gloo():
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    expect = 0.5 * (_replica_grads(0, 2) + _replica_grads(1, 2))
    for r in range(world):
        assert np.abs(out[r] - expect).max() <= 1e-12 * max(1.0, np.abs(expect).max())
    assert np.array_equal(out[0], out[1])    # every replica applies the same update
    expect_g = np.concatenate([np.full((3, 2, 4), float(r)) + np.arange(4.0) for r in range(world)], axis=1)
    for r in range(world):
        assert np.array_equal(out["gather%d" % r], expect_g)
        assert out["ragged%d" % r] == "refused"
