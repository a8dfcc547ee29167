"""One data-parallel replica of a small 3-modality job (run by tests/test_dp_gpu.py, one process per rank, gloo rendezvous
on 127.0.0.1 so that two replicas can share the one GPU of a test box; the launch under torchrun uses RCCL instead).

usage: dp_worker.py <mode: replica|global> <out.npz>      (RANK / WORLD_SIZE / MASTER_* from the environment)
Writes the all-reduced gradient (scaled as Adam would apply it), the losses and, after one optimizer step, the parameters.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist

from tests.synth import make_batch
from ugaitnet_amd import dp
from ugaitnet_amd.engine import GaitCore

KINDS, B, L, NCLS, IDS = ("of", "gray", "depth"), 8, 3, 6, 4


def job_batch():
    return make_batch(KINDS, B, L, NCLS, ids=IDS, seed=77)


def make_core(world, mode):
    return GaitCore([2, 1, 1], nclasses=NCLS, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=5, lr=1e-4,
                    world_size=world, dp_mode=mode)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    backend = os.environ.get("UGN_DP_BACKEND", "gloo")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    rank, world, _ = dp.init_from_env(backend=backend)
    xs, uses, labels, onehot = job_batch()
    lo, hi = dp.shard_bounds(B, rank, world)
    core = make_core(world, mode)
    cut = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).cuda()
    args = ([cut(x) for x in xs], [cut(u) for u in uses], labels[lo:hi], cut(onehot))
    core.forward_backward(*args)
    losses = core.losses()
    scale = core.finish_gradient_allreduce()      # bucketed, overlapped with the backward pass (UGN_AR_OVERLAP=0: one call)
    grad = (core.store.grad * scale).cpu().numpy()
    core.train_step(*args)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, grad=grad, loss=losses["loss"], triplet=losses["triplet"], xent=losses["xent"],
                 params=core.store.flat.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
