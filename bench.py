#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the 3-modality UGaitNet hot path, forward + backward + Adam, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], SURVEY.md section 8d "C3"): 3 modalities (optical flow 2ch + gray + depth),
25 frames of 60x60, 24 clips per GPU, 12 ids x 2, 150 classes, sign_max fusion with the 7-pattern missing-modality
masks, loss = 1.0*batch-all triplet(0.2) + 0.1*cross-entropy, Adam(1e-4).  Synthetic data, seed 232323, random-init
weights.  One process per GPU; for N > 1 every rank processes its own 24 clips (weak scaling), per-replica loss as the
reference's MirroredStrategy does, gradients averaged with one RCCL all-reduce over the flat gradient buffer.
Inputs are resident in HBM before the timed region starts.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KINDS = ("of", "gray", "depth")
B_PER_GPU, L, NCLS = 24, 25, 150
# secondary workloads (never the headline line): --workload c4 = SURVEY 8(d) "C4", the CASIA-B shape
WORKLOADS = {
    "c3": dict(kinds=("of", "gray", "depth"), clips=24, ncls=150, ids_per=2,
               text="C3: 3 modalities (of 2ch + gray + depth), 25x60x60, %d clips/GPU, %d ids x %d, 150 classes, sign_max, "
                    "7-pattern masks, triplet(0.2)+0.1*xent, Adam 1e-4"),
    "c4": dict(kinds=("of", "gray", "sil"), clips=40, ncls=74, ids_per=10,
               text="C4: 3 modalities (of 2ch + gray + silhouette), 25x60x60, %d clips/GPU, %d ids x %d, 74 classes, sign_max, "
                    "7-pattern masks, triplet(0.2)+0.1*xent, Adam 1e-4"),
    # C5 = C4 at 128 clips per 8-GPU node with bf16 MFMA operands: 16 clips per GPU; always run with --dtype bf16
    "c5": dict(kinds=("of", "gray", "sil"), clips=16, ncls=74, ids_per=8,
               text="C5: 3 modalities (of 2ch + gray + silhouette), 25x60x60, %d clips/GPU, %d ids x %d, 74 classes, sign_max, "
                    "7-pattern masks, triplet(0.2)+0.1*xent, Adam 1e-4, bf16 MFMA operands"),
}
# algorithmic FLOPs: forward per clip per modality, exact from the layer shapes
F_FWD = {1: 7.944e9, 2: 8.108e9}
F_FIRST = {1: 0.164e9, 2: 0.328e9}
FLOP_PER_CLIP = sum(3 * F_FWD[c] - F_FIRST[c] for c in (2, 1, 1))   # fwd + dgrad + wgrad, no dgrad for layer 1
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA = 16 * PEAK_F32_MFMA   # same guide: the f32 MFMA rate is 1/16 of the dense bf16 rate (~2.5 PFLOP/s)


def cpu_baseline(seconds_budget=25.0):
    """The CPU restatement (torch CPU ops, oneDNN) timed on this host on a bounded sample of the same workload."""
    import torch
    from oracle import torch_ref as T
    from oracle import ugaitnet_oracle as O
    from tests.synth import make_batch
    b = 8  # 4 ids x 2, the first 8 rows of the 7-pattern mask cycle
    xs, uses, labels, onehot = make_batch(KINDS, b, L, NCLS, ids=4, seed=232323)
    rng = np.random.default_rng(0)
    params = dict(branches=[O.init_branch_params(rng, 2 if k == "of" else 1) for k in KINDS],
                  head=O.init_head_params(rng, NCLS))
    tr = T.TorchTrainer(T.params_from_numpy(params), lr=1e-4, margin=0.2, loss_weights=(1.0, 0.1))
    txs = [torch.from_numpy(x) for x in xs]
    tus = [torch.from_numpy(u) for u in uses]
    tl, to = torch.from_numpy(labels), torch.from_numpy(onehot)
    tr.step(txs, tus, tl, to)  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start < seconds_budget and len(times) < 10):
        t0 = time.perf_counter()
        tr.step(txs, tus, tl, to)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=b / med, unit="clips/s", cores=int(torch.get_num_threads()), kind="port",
                sample="%d clips (same 3-modality shape, masks, L=25), %d timed steps after 1 warm-up, median; "
                       "torch-CPU (oneDNN) restatement oracle/torch_ref.py, fwd+bwd+Adam" % (b, len(times)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-only", action="store_true", help="skip the secondary mask-skipping run (profiling target)")
    ap.add_argument("--skip-masked", action="store_true",
                    help="run each encoder only on the clips whose modality flag is 1 (exactly the same results; the default "
                         "line computes the masked pairs too)")
    ap.add_argument("--dp-mode", choices=("replica", "global"), default="replica",
                    help="N > 1: 'replica' = the reference's MirroredStrategy (losses per replica slice, one gradient "
                         "all-reduce); 'global' = all-gather the fused features so the losses see the whole batch")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16 = BASELINE configs[4] / SURVEY C5 arithmetic: bf16 operands in the MFMA of the 3x3 forward "
                         "convolutions and data gradients, fp32 accumulate, fp32 tensors and weight gradients; never the headline")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3", help="c3 is the headline workload")
    ap.add_argument("--clips-per-gpu", type=int, default=0,
                    help="override the workload's batch (e.g. 96 = the generator-expanded C3 batch); 0 = the workload's own")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.workload == "c5":
        args.dtype = "bf16"
    global KINDS, B_PER_GPU, NCLS
    KINDS, NCLS = wl["kinds"], wl["ncls"]
    B_PER_GPU = args.clips_per_gpu or wl["clips"]
    if B_PER_GPU % wl["ids_per"]:
        raise SystemExit("--clips-per-gpu must be a multiple of %d for workload %s" % (wl["ids_per"], args.workload))
    n_ids = B_PER_GPU // wl["ids_per"]

    import torch
    import torch.distributed as dist
    from tests.synth import make_batch
    from ugaitnet_amd import ops
    from ugaitnet_amd.engine import GaitCore

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # UGN_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices); the real
    # launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("UGN_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    dev = torch.device("cuda", local)
    if world > 1:   # create the communicator now, so that a run with --warmup 0 does not time its construction
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()

    xs, uses, labels, onehot = make_batch(KINDS, B_PER_GPU, L, NCLS, ids=n_ids, seed=232323 + rank)
    core = GaitCore([2, 1, 1], nclasses=NCLS, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), device=dev,
                    seed=232323, lr=1e-4, world_size=world, skip_masked=args.skip_masked, dp_mode=args.dp_mode, conv_precision=args.dtype)
    dxs = [torch.from_numpy(x).to(dev) for x in xs]
    dus = uses if args.skip_masked else [torch.from_numpy(u).to(dev) for u in uses]   # flags: host copies when they steer the launch
    doh = torch.from_numpy(onehot).to(dev)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        core.train_step(dxs, dus, labels, doh)
    sync_all()
    ops.TIMING.clear()
    ops.TIMING_ENABLED = True   # HIP-event pairs around the dominant kernel's launches, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        core.train_step(dxs, dus, labels, doh)
    sync_all()
    dt = time.perf_counter() - t0
    ops.TIMING_ENABLED = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = core.losses()

    # secondary figure, same workload: encoders run only on the clips whose modality flag is 1 (the gate multiplies the
    # rest by 0, so every result is unchanged; tests/test_fullsize_gpu.py).  Never reported as `value`.
    skip_rate = None
    if not args.skip_masked and not args.dense_only:
        del core
        torch.cuda.empty_cache()
        core2 = GaitCore([2, 1, 1], nclasses=NCLS, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), device=dev,
                         seed=232323, lr=1e-4, world_size=world, skip_masked=True, dp_mode=args.dp_mode, conv_precision=args.dtype)
        for _ in range(args.warmup):
            core2.train_step(dxs, uses, labels, doh)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            core2.train_step(dxs, uses, labels, doh)
        sync_all()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        skip_rate = world * B_PER_GPU * args.steps / dt2

    if rank == 0:
        clips = world * B_PER_GPU * args.steps
        value = clips / dt
        # roofline of the dominant kernel: the 3x3 32->32 conv at 64x64 (layer a2), forward launch
        name, flops = ops.ROOFLINE_OP, ops.ROOFLINE_FLOPS_PER_FRAME
        evs = ops.TIMING.get(name, [])
        roof = None
        if evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            avg_s = float(np.mean(ms)) * 1e-3
            frames = B_PER_GPU * L
            achieved = flops * frames / avg_s / 1e12
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            peak = PEAK_F32_MFMA if args.dtype == "f32" else PEAK_BF16_MFMA
            roof = dict(bound="mfma", achieved=round(achieved, 2), peak=round(peak / 1e12, 1), unit="TFLOP/s",
                        frac=round(achieved * 1e12 / peak, 4), traffic=traffic if args.dtype == "f32" else None, kernel=name,
                        launches=len(evs), avg_us=round(avg_s * 1e6, 1),
                        note="achieved = direct-convolution (algorithmic) FLOPs / time; the kernel is Winograd F(2x2,3x3) and "
                             "executes 1/2.25 of them on the fp32 MFMA, so frac can exceed 1")
            others = []
            for oname, oflops in ops.EXTRA_TIMED.items():
                oev = ops.TIMING.get(oname, [])
                if oev:
                    oavg = float(np.mean([a.elapsed_time(b) for a, b in oev])) * 1e-3
                    otf = oflops * (frames + B_PER_GPU) / oavg / 1e12   # launched together with its set-level twin (B more images)
                    others.append(dict(kernel=oname, achieved=round(otf, 2), frac=round(otf * 1e12 / peak, 4),
                                       avg_us=round(oavg * 1e6, 1), launches=len(oev)))
            roof["other_kernels"] = others
        out = dict(metric="clips/sec (3-mod, L=25, 60x60) fwd+bwd+Adam", value=round(value, 2), unit="clips/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32" if args.dtype == "f32" else "bf16 MFMA operands in the 3x3 fwd/dgrad (f32 tensors, accumulate, wgrad)",
                   data="synthetic",
                   config=dict(workload=wl["text"] % (B_PER_GPU, n_ids, wl["ids_per"]),
                               clips_per_gpu=B_PER_GPU, parallelism="dp%d" % world, dp_mode=args.dp_mode, masked_pairs_skipped=bool(args.skip_masked)),
                   whole_step_tflops=round(value * FLOP_PER_CLIP / 1e12, 2),
                   whole_step_frac_of_f32_mfma_peak=round(value * FLOP_PER_CLIP / world / PEAK_F32_MFMA, 4),   # (always the f32 peak)
                   loss=round(losses["loss"], 5), roofline=roof)
        if skip_rate is not None:
            out["value_skip_masked"] = round(skip_rate, 2)   # 29 of the 72 (clip, modality) pairs of this batch are masked
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
