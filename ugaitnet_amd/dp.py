"""Data-parallel plumbing: one process per GPU, RCCL over xGMI via torch.distributed (backend "nccl" IS RCCL on ROCm).

Replaces the reference's `tf.distribute.MirroredStrategy()` scope (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:342-349,
458-461): every replica owns a contiguous slice of the batch, evaluates the loss on ITS slice (Keras per-replica loss,
scaled by 1/replicas) and the parameter gradients are summed across replicas.  The only exchange on the path is that
gradient all-reduce over the flat fp32 gradient buffer (35.8-40.7 MB): one collective after the
backward pass, or with UGN_AR_OVERLAP=1 four buckets (head, then one per branch) issued as the backward pass produces them
(`allreduce_sum_async`, GaitCore._reduce_bucket); the 1/world factor is folded into the Adam kernel.

Global-batch mode (`GaitCore(dp_mode="global")`, SURVEY.md section 8e collective (1)): the two places where samples are
coupled -- the batch-axis l2_normalize (nets/mj_uwyhNets_ba.py:817,1191) and the triplet loss
(nets/triplet_loss_all.py:8-77) -- are evaluated on the batch of ALL replicas, so that G GPUs x B/G clips compute exactly
what one device computes on B clips.  That costs one extra all-gather of the fused features [62, B/G, 256]
(`gather_batch_axis`, 0.3-1 MB per rank) and the gradients are then summed, not averaged.

These helpers are backend-agnostic so that the logic is testable with gloo on CPU.
"""
from __future__ import annotations

import os

import numpy as np


# bench.py sets TIMING = {} when ranks > 1 (or --force-dist): every collective of the path is then bracketed by an event pair on
# the stream it is issued on, and the JSON line reports milliseconds per step per collective kind.
TIMING = None


def _timed(name, fn):
    if TIMING is None:
        return fn()
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    TIMING.setdefault(name, []).append((e0, e1))
    return r


def timing_summary(steps):
    """{collective: ms per step} from the recorded event pairs (synchronises)."""
    import torch
    if not TIMING:
        return {}
    torch.cuda.synchronize()
    return {k: round(sum(a.elapsed_time(b) for a, b in v) / max(1, steps), 4) for k, v in TIMING.items()}


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend="nccl"):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of a batch of n rows owned by `rank` (remainder rows go to the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(X, y, rank, world):
    """Slice every array of the generator's batch (X list, y list) along axis 0."""
    n = np.asarray(X[0] if isinstance(X, (list, tuple)) else X).shape[0]
    lo, hi = shard_bounds(n, rank, world)
    cut = lambda a: a[lo:hi]
    Xs = [cut(a) for a in X] if isinstance(X, (list, tuple)) else cut(X)
    ys = [cut(a) for a in y] if isinstance(y, (list, tuple)) else cut(y)
    return Xs, ys


def allreduce_sum_(flat, group=None, force=False):
    """In-place SUM all-reduce of the flat gradient buffer; returns the factor that turns the sum into the
    replica mean (applied inside ugn_adam_step).  `force`: issue the collective even in a one-rank group (rehearsal of the
    RCCL path on a one-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return 1.0
    _timed("allreduce_grad_ms", lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group))
    return 1.0 / world


def gather_batch_axis(t, axis, group=None, check=True, force=False):
    """All-gather `t` over the replicas and concatenate along `axis` in rank order.  Every replica must hold the same
    shape (equal slices of the global batch); with check=True a mismatch is an error on every rank, not a hang (the
    comparison synchronises with the host: the engine checks on the label exchange, before the step's kernels are queued,
    and gathers the features of the same step unchecked)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return t
    world = dist.get_world_size(group)
    if check:
        shape = torch.tensor(list(t.shape), dtype=torch.int64, device=t.device)
        shapes = [torch.empty_like(shape) for _ in range(world)]
        dist.all_gather(shapes, shape, group=group)
        if any(not torch.equal(s, shape) for s in shapes):
            raise ValueError("global-batch mode needs equal per-replica shapes, got %r" % ([s.tolist() for s in shapes],))
    parts = [torch.empty_like(t) for _ in range(world)]
    _timed("allgather_ms", lambda: dist.all_gather(parts, t.contiguous(), group=group))
    return torch.cat(parts, dim=axis)


def group_rank(group=None):
    import torch.distributed as dist
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_async(t, group=None, force=False):
    """SUM all-reduce of `t` (a slice of the flat gradient buffer) issued behind the CURRENT stream's work and left running
    on the communication stream; returns the work handle (`.wait()` orders the caller's stream after it) or None for a single
    process."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
