#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the 3-modality UGaitNet hot path, forward + backward + Adam, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], SURVEY.md section 8d "C3"): 3 modalities (optical flow 2ch + gray + depth),
25 frames of 60x60, 24 clips per GPU, 12 ids x 2, 150 classes, sign_max fusion with the 7-pattern missing-modality
masks, loss = 1.0*batch-all triplet(0.2) + 0.1*cross-entropy, Adam(1e-4).  Synthetic data, seed 232323, random-init
weights.  Inputs are resident in HBM before the timed region starts.

N > 1: one process per GPU over RCCL (torch.distributed backend "nccl").  Started by the driver's torchrun line, or --
when WORLD_SIZE is not in the environment -- by this script itself, which then spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as CHILD processes before anything touches the GPU, relays
rank 0's JSON line and exits with the children's code.  Default = weak scaling: every rank processes its own 24 clips with
per-replica losses as the reference's MirroredStrategy does (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:342-349), gradients
averaged by one all-reduce of the flat gradient buffer.  `--scaling strong` = the reference's own sharding of a fixed batch
(mains/...CasiaB.py:458-461: batchsize / multigpu): the workload's clips are split over the N ranks, C4 = 40 / N per GPU.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L = 25
# secondary workloads (never the headline line): --workload c4 = SURVEY 8(d) "C4", the CASIA-B shape
WORKLOADS = {
    # C2 = BASELINE.json configs[1]: BL-single gray, the single-modality graph (no gate, no normalisation:
    # nets/mj_uwyhNets_ba.py:893-903), 24 clips = 12 ids x 2, 150 classes
    "c2": dict(kinds=("gray",), clips=24, ncls=150, ids_per=2, multimodal=False,
               text="C2: BL-single gray (single-modality graph), 25x60x60, %d clips/GPU, %d ids x %d, 150 classes, "
                    "triplet(0.2)+0.1*xent, Adam 1e-4"),
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


def flop_per_clip(kinds):
    return sum(3 * F_FWD[2 if k == "of" else 1] - F_FIRST[2 if k == "of" else 1] for k in kinds)


DTYPE_TEXT = {
    "f32x3": "f32: IEEE fp32 tensors in HBM end to end (no storage format, no block exponent); the 3x3 layers multiply them on "
             "v_mfma_f32_16x16x32_bf16 through the EXACT three-way bf16 split of both operands (x = x0 + x1 + x2, 24 = 8 + 8 + 8 bits, "
             "fp32's exponent range), six of the nine partial products per fp32 product (the dropped ones < 2^-23 of the product: on the "
             "hardware all nine give the same error against fp64 to three digits), fp32 accumulate: error against fp64 at the level of "
             "the fp32-MFMA kernels of the same library on the same inputs (tests/test_x3_gpu.py, tools/x3_accuracy.py); 5x5 layer, "
             "pooling, head, losses, Adam in fp32",
    "f32": "f32: IEEE fp32 tensors and arithmetic end to end, 3x3 layers as Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32",
    "bf16": "bf16: activations, gradients and saved tensors bf16 in HBM, 3x3 layers on v_mfma_f32_32x32x16_bf16 with f32 accumulate, "
            "f32 weight gradients / master weights / Adam (BASELINE configs[4])",
    "h2": "f16x2: fp32-class values held as two f16 halves + a block exponent (22 significant bits); 3x3 layers = three "
          "v_mfma_f32_16x16x32_f16 per product (hi*hi + hi*lo + lo*hi; the MaxPool'ed layers' weight gradients: v_smfmac_f32_16x16x64_f16, "
          "the pooled gradient being 2:4 sparse along a pixel row), f32 accumulate; everything else f32",
}
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
PEAK_BF16_MFMA = 16 * PEAK_F32_MFMA   # same guide: the f32 MFMA rate is 1/16 of the dense bf16 rate (~2.5 PFLOP/s)
PEAK_HBM = 8.0e12          # same guide: HBM3E spec peak (6.3 TB/s is what a float4 copy achieves)
DEFAULT_DTYPE = os.environ.get("UGN_BENCH_DTYPE", "f32x3")     # the product's default arithmetic (engine.DEFAULT_PRECISION)
# the SAME job in the library's other fp32-class arithmetics, timed after the headline run and reported beside it (never as `value`):
# key suffix -> conv_precision
# (the f16x2 set is an opt-in build since round 6 -- python -m ugaitnet_amd.build --h2 -- and carries no credit: it is timed beside the
#  headline only when --with-h2-line asks for it and the library has it)
SECONDARY = {"f32x3": (("f32_mfma", "f32"),), "f32": (), "h2": (("f32_mfma", "f32"),), "bf16": ()}
EXEC_FACTOR = {"f32x3": 6.0, "f32": 16.0 / 36.0, "bf16": 1.0, "h2": 3.0}


def cpu_baseline(kinds, ncls, clips, n_ids, seconds_budget=30.0, multimodal=True):
    """The CPU restatement (torch CPU ops, oneDNN) timed on this host on a bounded sample of the same workload: the SAME
    batch shape (all clips of one step), one warm-up on a small batch (pages in oneDNN's kernels), then whole steps until the
    budget is spent (at least one)."""
    import torch
    from oracle import torch_ref as T
    from oracle import ugaitnet_oracle as O
    from tests.cpu_share import usable_cores
    from tests.synth import make_batch
    # the cores this process may REALLY use (scheduler affinity and the container's CPU quota, not the machine's logical CPUs: a GPU
    # box reports 128 and grants 16): the thread pool is sized to them and `cores` reports them
    cores = usable_cores()
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    params = dict(branches=[O.init_branch_params(rng, 2 if k == "of" else 1) for k in kinds],
                  head=O.init_head_params(rng, ncls))
    tr = T.TorchTrainer(T.params_from_numpy(params), lr=1e-4, margin=0.2, loss_weights=(1.0, 0.1), multimodal=multimodal)

    def tensors(b, ids):
        xs, uses, labels, onehot = make_batch(kinds, b, L, ncls, ids=ids, seed=232323)
        return ([torch.from_numpy(x) for x in xs], [torch.from_numpy(u) for u in uses] if multimodal else None,
                torch.from_numpy(labels), torch.from_numpy(onehot))
    tr.step(*tensors(4, 2))  # warm-up
    batch = tensors(clips, n_ids)
    times = []
    t_start = time.perf_counter()
    while not times or (time.perf_counter() - t_start + times[-1] < seconds_budget and len(times) < 5):
        t0 = time.perf_counter()
        tr.step(*batch)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=clips / med, unit="clips/s", cores=int(cores), logical_cpus_of_the_machine=int(os.cpu_count() or 0), kind="port",
                sample="%d whole step(s) of the same %d-clip batch (%d modalit%s, L=25) after a 4-clip warm-up, median; "
                       "torch-CPU (oneDNN) restatement oracle/torch_ref.py, fwd+bwd+Adam; the TF-2.3 reference cannot run here"
                       % (len(times), clips, len(kinds), "ies, masks" if multimodal else "y"))


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-only", action="store_true",
                    help="skip the secondary runs (mask-skipping, IEEE-fp32 line): only the named arithmetic runs (profiling target)")
    ap.add_argument("--no-secondary-lines", "--no-f32-line", dest="no_f32_line", action="store_true",
                    help="headline arithmetic only: do not time the same job in the other fp32-class arithmetics beside it "
                         "(value_f32_mfma = Winograd on the fp32 MFMA, value_h2 = f16x2 tensors)")
    ap.add_argument("--with-h2-line", action="store_true",
                    help="also time the same job in the opt-in f16x2 arithmetic (value_h2); needs a library built with --h2")
    ap.add_argument("--skip-masked", action="store_true",
                    help="run each encoder only on the clips whose modality flag is 1 (exactly the same results; the default "
                         "line computes the masked pairs too)")
    ap.add_argument("--dp-mode", choices=("replica", "global"), default=None,
                    help="N > 1: 'replica' = the reference's MirroredStrategy (losses per replica slice, one gradient "
                         "all-reduce; default for weak scaling); 'global' = all-gather the fused features so the losses see the "
                         "whole batch (default for --scaling strong: N GPUs then compute the one-GPU step)")
    ap.add_argument("--dtype", choices=("f32x3", "f32", "bf16", "h2"), default=DEFAULT_DTYPE,
                    help="f32x3 (default) = fp32 tensors, 3x3 products through the exact three-way bf16 split on the bf16 matrix pipe; "
                         "f32 = fp32 tensors, Winograd on the fp32 MFMA; h2 = fp32-class values stored as two f16 halves + a block "
                         "exponent (22 bits), 3x3 layers on the f16 matrix pipe; bf16 = BASELINE configs[4] arithmetic, never the headline")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3", help="c3 is the headline workload")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="strong: the workload's batch is split over the ranks (c4: 40 / N clips per GPU)")
    ap.add_argument("--clips-per-gpu", type=int, default=0,
                    help="override the per-GPU batch (e.g. 96 = the generator-expanded C3 batch; 5 = one rank's share of C4 on 8 "
                         "GPUs); 0 = the workload's own")
    ap.add_argument("--serial", action="store_true",
                    help="run the timed region itself with every launch on one stream (the rocprofv3 kernel-trace target: "
                         "per-kernel durations are then clean); never the headline")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the serialised per-kernel timing pass")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as captured hipGraphs (engine.GraphedTrainStep: same kernels, no per-launch host cost); "
                         "for host-bound set-ups.  Measured on one MI355X box: NOT faster -- 5 clips 2.13 ms against 1.96 eager, 24 "
                         "clips 6.56 against 6.35 (the replay has one forward chain and no repack beside the 5x5 layer)")
    ap.add_argument("--kernel-table", default="", help="write every row of the serialised per-kernel pass to this CSV file")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the gradient collectives even with one rank (rehearses the RCCL "
                         "code path on a one-GPU box)")
    return ap


def launch_ranks(args, argv):
    """--gpus N > 1 without a torchrun environment: start the N ranks as children (never exec), relay rank 0's line."""
    import torch   # device_count() does not initialise the GPU on this image
    backend = os.environ.get("UGN_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; one rank per GPU over RCCL needs %d "
                         "(UGN_DIST_BACKEND=gloo rehearses more ranks than GPUs)" % (args.gpus, have, args.gpus))
    if have < 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)   # (stderr passes through)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print("bench.py: the %d-rank launch failed (exit code %d, %d JSON lines)" % (args.gpus, r.returncode, len(lines)),
              file=sys.stderr)
        raise SystemExit(r.returncode or 1)
    print(lines[0], flush=True)


def roofline_pass(core, batch, steps, dtype, table_path="", step_ms=None):
    """Per-kernel durations from a fully serialised pass (one stream, HIP-event pair on that stream around every launch),
    taken AFTER the timed region; returns the roofline object of the kernel with the largest total duration."""
    import torch
    from ugaitnet_amd import _lib, engine
    dev = core.device
    # The event pairs sit on the stream around each launch, so a pair measures the kernel only while the stream is never EMPTY when the
    # pair is queued: with the GPU ahead of the host (small batches: a launch takes the host ~20 us to queue with its two events, many of
    # the step's kernels less to run) the first event completes at once and the pair then contains the HOST's time to issue the launch --
    # a host hiccup (GC, a page fault, a neighbour on the box's cores) lands inside the bracket.  That is what VERDICT r05 item 2's
    # "stalled launch" was (test_roofline_object[f32], 8 clips per GPU: frac 0.0091 = one ~50-us Winograd launch measured at ~60x), a
    # measurement bug, not a device stall.  Round 6: every profiled step is queued BEHIND A GATE -- a spin kernel (torch.cuda._sleep)
    # long enough for the host to queue everything behind it, then plain steps of the same workload (the clocks of the timed region) --
    # so all pairs are device-side back-to-back stamps.  The filter stays as a second line of defence, and what it drops is recorded
    # (label, position, duration, the label queued before it, whether it was that label's first launch of the pass) instead of counted.
    host_ms, host_plain_ms, gate_cycles, warm = 0.0, 0.0, 0, 0
    with core.serial_launches():
        core.train_step(*batch)            # un-timed: first step on the one-stream schedule
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        core.train_step(*batch)            # what the host needs to queue one step WITHOUT event pairs ...
        host_plain_ms = (time.perf_counter() - t0) * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        prof, work, order = {}, {}, []
        try:
            _lib.PROFILE, _lib.WORK = {}, {}
            e0.record()
            t0 = time.perf_counter()
            core.train_step(*batch)            # ... and WITH them (un-timed as well)
            host_ms = (time.perf_counter() - t0) * 1e3
            e1.record()
            torch.cuda.synchronize()
            step_ms = max(e0.elapsed_time(e1), 0.05) if step_ms is None else step_ms
            # the gate of one profiled step: a spin kernel that covers the host's queueing of everything behind it, then `warm` plain
            # steps (>= 50 ms of the real workload: behind 5-10 ms of a near-idle spin the chip's clocks are not those of the timed region
            # -- the first gated build of this pass measured every convolution 15-25 % slower than the step's own kernels, with 16 ms of
            # plain steps in between still 3-6 % slower than rocprofv3's back-to-back run of the same kernels), then the profiled step
            warm = max(2, int(np.ceil(50.0 / step_ms)))
            import torch.distributed as _dist
            if _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1:
                # every step carries the gradient all-reduce: all ranks must run the SAME number of steps in this pass (a rank's own
                # timing must not decide it)
                wt = torch.tensor([warm], device=dev, dtype=torch.int32)
                _dist.all_reduce(wt, op=_dist.ReduceOp.MAX)
                warm = int(wt.item())
            spin_ms = min(1.5 * (warm * host_plain_ms + host_ms) + 2.0, 400.0)
            gate_cycles = int(spin_ms * 1e-3 * 2.4e9)      # (spin cycles at <= 2.4 GHz)
            _lib.WORK, _lib.ORDER = work, order
            for _ in range(steps):
                _lib.PROFILE = None
                torch.cuda._sleep(gate_cycles)
                for _w in range(warm):
                    core.train_step(*batch)
                _lib.PROFILE = prof
                core.train_step(*batch)
                _lib.PROFILE = None
                torch.cuda.synchronize()
        finally:
            _lib.PROFILE, _lib.ORDER = None, None
    rows, dropped = [], []
    prev_of, first_of, count = {}, {}, {}
    for pos, label in enumerate(order):         # (label, k-th launch of that label) -> the label queued before it, first of its label?
        k = count.get(label, 0)
        count[label] = k + 1
        prev_of[(label, k)] = order[pos - 1] if pos else None
        first_of[(label, k)] = k == 0
    for label, evs in prof.items():
        us = [a.elapsed_time(b) * 1e3 for a, b in evs]
        # a launch beyond five times its label's median is left out of the label's average -- and RECORDED
        med = float(np.median(us))
        keep = [u for u in us if u <= 5.0 * med] or us
        for k, u in enumerate(us):
            if u > 5.0 * med and len(keep) < len(us):
                dropped.append(dict(label=label, index_of_label=k, us=round(u, 1), median_us=round(med, 1),
                                    queued_after=prev_of.get((label, k)), first_launch_of_label_in_pass=bool(first_of.get((label, k), False))))
        rows.append((float(np.mean(keep)) * len(us), float(np.mean(keep)), len(us), label))
    stalled = len(dropped)
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)

    def describe(row):
        tot, avg, n, label = row
        w = work.get(label)
        d = dict(kernel=label, launches_per_step=n // steps, avg_us=round(avg, 1), share_of_step=round(tot / total, 4))
        if w and w["bound"] == "roof":
            # both roofs (f16x2 3x3 kernels): executed matrix FLOPs against the dense f16 peak AND algorithmic bytes against 8 TB/s;
            # `bound` = the roof whose floor (time at peak) is higher, and achieved / peak / unit / frac are that roof's
            t = avg * 1e-6
            mfma_frac, hbm_frac = w["mfma_flops"] / t / PEAK_BF16_MFMA, w["bytes"] / t / PEAK_HBM
            mfma_floor, hbm_floor = w["mfma_flops"] / PEAK_BF16_MFMA, w["bytes"] / PEAK_HBM
            if mfma_floor >= hbm_floor:
                d.update(bound="mfma", unit="TFLOP/s", peak=round(PEAK_BF16_MFMA / 1e12, 1), achieved=round(w["mfma_flops"] / avg / 1e6, 2),
                         frac=round(mfma_frac, 4))
            else:
                d.update(bound="hbm", unit="GB/s", peak=PEAK_HBM / 1e9, achieved=round(w["bytes"] / avg / 1e3, 1), frac=round(hbm_frac, 4))
            d.update(mfma_frac=round(mfma_frac, 4), hbm_frac=round(hbm_frac, 4), mfma_floor_us=round(mfma_floor * 1e6, 1),
                     hbm_floor_us=round(hbm_floor * 1e6, 1), mfma_tflops=round(w["mfma_flops"] / avg / 1e6, 2),
                     algorithmic_tflops=round(w["flops"] / avg / 1e6, 2), algorithmic_gbytes_per_s=round(w["bytes"] / avg / 1e3, 1),
                     rocprof_kernel=w["kernel"], images_per_launch=w["images"])
            if w.get("dtype") == "bf16x3":
                # the contract's literal reading for an fp32 path -- ALGORITHMIC FLOPs against the dense MFMA peak of the dtype the path
                # computes in (fp32: 157.3 TFLOP/s) -- beside `frac`: above 1 means the launch outruns what ANY kernel issuing fp32
                # matrix instructions could reach, which is what the three-way bf16 split is for
                d["algorithmic_frac_of_f32_mfma_peak"] = round(w["flops"] / t / PEAK_F32_MFMA, 4)
        elif w and w["bound"] == "mfma":
            peak = PEAK_BF16_MFMA if w["dtype"] in ("bf16", "f16x2") else PEAK_F32_MFMA   # (f16 MFMAs run at the bf16 rate)
            d.update(bound="mfma", unit="TFLOP/s", peak=round(peak / 1e12, 1), achieved=round(w["mfma_flops"] / avg / 1e6, 2),
                     frac=round(w["mfma_flops"] / (avg * 1e-6) / peak, 4), algorithmic_tflops=round(w["flops"] / avg / 1e6, 2),
                     rocprof_kernel=w["kernel"], images_per_launch=w["images"])
        elif w and w["bound"] == "hbm":
            d.update(bound="hbm", unit="GB/s", peak=PEAK_HBM / 1e9, achieved=round(w["bytes"] / avg / 1e3, 1),
                     frac=round(w["bytes"] / (avg * 1e-6) / PEAK_HBM, 4), rocprof_kernel=w["kernel"])
            if w.get("mfma_flops"):      # (bf16 3x3 layers: HBM first, the matrix pipe second)
                d.update(mfma_tflops=round(w["mfma_flops"] / avg / 1e6, 2), mfma_frac=round(w["mfma_flops"] / (avg * 1e-6) / PEAK_BF16_MFMA, 4),
                         images_per_launch=w.get("images"))
        return d
    if table_path:
        with open(table_path, "w") as f:
            f.write("label,launches_per_step,avg_us,total_us_per_step,share,bound,achieved,peak,unit,frac,mfma_frac,hbm_frac,rocprof_kernel\n")
            for row in rows:
                d = describe(row)
                f.write('"%s",%d,%.1f,%.1f,%.4f,%s,%s,%s,%s,%s,%s,%s,"%s"\n' % (
                    d["kernel"], d["launches_per_step"], d["avg_us"], row[0] / steps, d["share_of_step"], d.get("bound", ""),
                    d.get("achieved", ""), d.get("peak", ""), d.get("unit", ""), d.get("frac", ""), d.get("mfma_frac", ""),
                    d.get("hbm_frac", ""), d.get("rocprof_kernel", "")))
            for r in dropped:     # (launches left out of the averages above: beyond 5x their label's median)
                f.write('# stalled launch: "%s" launch %d of its label took %.1f us (median %.1f), queued after "%s", first of its label in the pass: %s\n'
                        % (r["label"], r["index_of_label"], r["us"], r["median_us"], r["queued_after"], r["first_launch_of_label_in_pass"]))
    top = describe(rows[0])
    roof = dict(bound=top.get("bound"), achieved=top.get("achieved"), peak=top.get("peak"), unit=top.get("unit"),
                frac=top.get("frac"), traffic=None)
    roof.update({k: v for k, v in top.items() if k not in roof})
    roof["how"] = ("dominant kernel = largest total duration in a serialised pass of %d steps after the timed region (every "
                   "launch on one stream, HIP-event pairs on that stream); achieved = FLOPs EXECUTED on the matrix pipe / avg "
                   "duration, against the peak of the instruction that executes them: f16x2 kernels run three f16 MFMAs per "
                   "fp32-class product (3x the direct-convolution count `algorithmic_tflops`, dense f16/bf16 peak); Winograd "
                   "F(2x2,3x3) fp32 kernels 16/36 of it (fp32-MFMA peak); HBM-bound kernels (and every bf16 3x3 kernel: mfma_frac "
                   "is their second figure): algorithmic bytes / avg duration against 8 TB/s.  x3 and f16x2 3x3 kernels carry BOTH figures "
                   "(mfma_frac, hbm_frac) and `bound` names the roof whose floor -- executed FLOPs / 2516.8 TFLOP/s or algorithmic "
                   "bytes / 8 TB/s -- is the longer time" % steps)
    # `traffic` is NOT measured in this run: it is the PMC figure (FETCH_SIZE x2-corrected + WRITE_SIZE, separate --pmc passes) of
    # the same kernel and launch size recorded by tools/profile_run.sh; absent (null) when no record matches
    for tname in ("roofline_traffic_%s.json" % dtype, "roofline_traffic.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tfile):
            continue
        t = json.load(open(tfile))
        hit = [r for r in t.get("kernels", [t]) if r.get("rocprof_kernel") == top.get("rocprof_kernel") and
               r.get("images_per_launch") == top.get("images_per_launch")]
        if hit:
            roof["traffic"] = hit[0].get("hbm_bytes_per_launch")
            roof["traffic_source"] = "profiles/" + tname + " (rocprofv3 --pmc passes of this kernel at this launch size; not this run)"
            break
    roof["serial_step_us"] = round(total / steps, 1)
    roof["stalled_launches"] = stalled          # launches beyond 5x their label's median (left out of the averages)
    if dropped:
        roof["stalled_launch_records"] = dropped[:16]
    roof["gate"] = dict(host_ms_per_profiled_step=round(host_ms, 2), host_ms_per_plain_step=round(host_plain_ms, 2),
                        spin_ms=round(gate_cycles / 2.4e6, 2), warm_steps_behind_the_spin=warm,
                        why="every profiled step is queued behind a spin kernel (covers the host's queueing) and >= 50 ms of plain steps "
                            "(the chip's clocks are those of the timed region), so no event pair contains host time")
    roof["other_kernels"] = [describe(r) for r in rows[1:10]]
    return roof


def run(args):
    wl = WORKLOADS[args.workload]
    if args.workload == "c5":
        args.dtype = "bf16"
    kinds, ncls = wl["kinds"], wl["ncls"]
    multimodal = wl.get("multimodal", True)

    import torch
    import torch.distributed as dist
    from tests.synth import make_batch
    from ugaitnet_amd import engine
    from ugaitnet_amd.engine import GaitCore

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.scaling == "strong":
        if args.clips_per_gpu:
            raise SystemExit("--scaling strong splits the workload's batch; --clips-per-gpu does not apply")
        if wl["clips"] % world:
            raise SystemExit("--scaling strong: %d clips do not split over %d ranks" % (wl["clips"], world))
        b_gpu = wl["clips"] // world
    else:
        b_gpu = args.clips_per_gpu or wl["clips"]
    dp_mode = args.dp_mode or ("global" if args.scaling == "strong" and world > 1 else "replica")
    # UGN_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices); the real
    # launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("UGN_DIST_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s): RCCL needs one GPU per rank" % (world, torch.cuda.device_count()))
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:     # --force-dist outside torchrun: a private one-rank rendezvous
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
            s.close()
        kw = dict(device_id=dev) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        dist.all_reduce(torch.zeros(1, device=dev))   # create the communicator now: --warmup 0 must not time its construction
        torch.cuda.synchronize()

    if args.scaling == "strong":
        # ONE global batch, every rank owns a contiguous slice of it (dp.shard_bounds), as MirroredStrategy splits a batch
        from ugaitnet_amd import dp
        gx, gu, glab, goh = make_batch(kinds, wl["clips"], L, ncls, ids=wl["clips"] // wl["ids_per"], seed=232323)
        lo, hi = dp.shard_bounds(wl["clips"], rank, world)
        xs, uses, labels, onehot = [x[lo:hi] for x in gx], [u[lo:hi] for u in gu], glab[lo:hi], goh[lo:hi]
        n_ids = wl["clips"] // wl["ids_per"]
    else:
        if b_gpu % wl["ids_per"]:
            ids_per = 1 if b_gpu < wl["ids_per"] else None
            if ids_per is None:
                raise SystemExit("--clips-per-gpu must be a multiple of %d for workload %s" % (wl["ids_per"], args.workload))
            n_ids = b_gpu          # fewer clips than one identity's share: every clip its own identity
        else:
            n_ids = b_gpu // wl["ids_per"]
        xs, uses, labels, onehot = make_batch(kinds, b_gpu, L, ncls, ids=n_ids, seed=232323 + rank)

    def make_core(skip, precision=None):
        return GaitCore([2 if k == "of" else 1 for k in kinds], nclasses=ncls, multimodal=multimodal, fuse_mode="sign_max", margin=0.2,
                        loss_weights=(1.0, 0.1), device=dev, seed=232323, lr=1e-4, world_size=world, skip_masked=skip and multimodal,
                        dp_mode=dp_mode, conv_precision=precision or args.dtype, force_collectives=args.force_dist)

    core = make_core(args.skip_masked)
    core_cfg = dict(ar_overlap=core.cfg.ar_overlap)
    dxs = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]
    dus_dev = [torch.from_numpy(np.ascontiguousarray(u)).to(dev) for u in uses] if multimodal else None
    doh = torch.from_numpy(np.ascontiguousarray(onehot)).to(dev)
    # flags: host copies when they steer the launch
    batch = (dxs, (uses if args.skip_masked else dus_dev) if multimodal else None, labels, doh)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(c, bt):
        step = c.train_step
        if args.graph and not c.skip_masked:       # (the mask-skipping secondary run launches by the batch's masks: eager)
            step = engine.GraphedTrainStep(c, *bt).step
        for _ in range(args.warmup):
            step(*bt)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(*bt)
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    ctx = core.serial_launches() if args.serial else None
    if ctx:
        ctx.__enter__()
    if use_dist:      # per-collective event pairs during the timed steps (warm-up included: divided by all steps run)
        from ugaitnet_amd import dp as _dp
        _dp.TIMING = {}
    dt = timed(core, batch)
    coll_ms = None
    if use_dist:
        # per-collective milliseconds per step from EVERY rank's view: each rank sums its own event pairs, the line carries the maximum
        # over ranks (what bounds the step) and the minimum (how far the ranks' views are apart)
        mine = _dp.timing_summary(args.steps + args.warmup)
        _dp.TIMING = None
        keys = sorted(mine)
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            keys = sorted(set().union(*[set(g) for g in gathered]))
            coll_ms = {k: dict(max_over_ranks=max(g.get(k, 0.0) for g in gathered), min_over_ranks=min(g.get(k, 0.0) for g in gathered))
                       for k in keys}
        else:
            coll_ms = {k: dict(max_over_ranks=mine[k], min_over_ranks=mine[k]) for k in keys}
    if ctx:
        ctx.__exit__(None, None, None)
    losses = core.losses()
    grad_bytes = int(core.store.numel * 4)

    roof = None
    if not args.no_roofline_pass:
        roof = roofline_pass(core, batch, 3, args.dtype, args.kernel_table if rank == 0 else "")

    # secondary figure, same workload: encoders run only on the clips whose modality flag is 1 (the gate multiplies the
    # rest by 0, so every result is unchanged; tests/test_fullsize_gpu.py).  Never reported as `value`.
    skip_rate = None
    if not args.skip_masked and not args.dense_only and not args.serial and multimodal:
        del core
        core = None
        torch.cuda.empty_cache()
        core2 = make_core(True)
        dt2 = timed(core2, (dxs, uses, labels, doh))
        skip_rate = world * b_gpu * args.steps / dt2
        del core2

    # secondary figure, same shapes: EVERY modality present in every clip (all `use` flags 1, no constant 1e-9 placeholder tensors) --
    # the dense rate on data with no constant frames (29 of C3's 72 (clip, modality) pairs are placeholders, and constant frames switch
    # fewer bits: the kernels hold a higher clock on them), which is also what every rank sees on batches without modality dropping
    # (nets/mj_uwyhNets_ba.py:1163-1180 with all flags 1).  Same steps / warm-up / barriers; its own dominant-kernel fraction.
    all_present = None
    if not (args.dense_only or args.serial or args.skip_masked or args.graph) and multimodal:
        core = None
        torch.cuda.empty_cache()
        from tests.synth import make_batch as _mb
        if args.scaling == "strong":
            fx = _mb(kinds, wl["clips"], L, ncls, ids=wl["clips"] // wl["ids_per"], seed=232323, all_present=True)
            fxs, fus = [x[lo:hi] for x in fx[0]], [u[lo:hi] for u in fx[1]]
        else:
            fxs, fus, _, _ = _mb(kinds, b_gpu, L, ncls, ids=n_ids, seed=232323 + rank, all_present=True)
        fb = ([torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in fxs],
              [torch.from_numpy(np.ascontiguousarray(u)).to(dev) for u in fus], labels, doh)
        c3 = make_core(False)
        dt3 = timed(c3, fb)
        roof3 = None if args.no_roofline_pass else roofline_pass(c3, fb, 3, args.dtype)
        all_present = dict(dt=dt3, loss=c3.losses()["loss"], roof=roof3)
        del c3, fb

    # The SAME job -- batch, initial weights, steps, warm-up, barriers -- in the library's other fp32-class arithmetics, reported BESIDE
    # the headline line (value_<tag>, ms_per_step_<tag>, roofline_<tag>, loss_<tag>), never as `value`.
    others = []
    if not (args.no_f32_line or args.dense_only or args.serial or args.skip_masked or args.graph):
        from ugaitnet_amd import _lib as _l
        secondary = list(SECONDARY[args.dtype])
        if args.with_h2_line and args.dtype != "h2":
            _l.require_h2("--with-h2-line")
            secondary.append(("h2", "h2"))
        for tag, prec in secondary:
            core = None
            torch.cuda.empty_cache()
            c2 = make_core(False, prec)
            dt2_ = timed(c2, batch)
            loss2 = c2.losses()["loss"]
            roof2 = None if args.no_roofline_pass else roofline_pass(c2, batch, 3, prec)
            others.append(dict(tag=tag, prec=prec, dt=dt2_, loss=loss2, roof=roof2))
            del c2

    if rank == 0:
        value = world * b_gpu * args.steps / dt
        dist_info = None
        if use_dist:
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
            from ugaitnet_amd import ops as _ops
            dist_info = dict(backend=dist.get_backend(), world_size=dist.get_world_size(), rccl_version=rccl,
                             allreduce=("bucketed (head + one per branch), issued as the backward pass produces them, RCCL's stream beside "
                                        "the remaining backward launches" if core_cfg["ar_overlap"] else "one call over the flat gradient buffer after the backward pass"),
                             ar_overlap=bool(core_cfg["ar_overlap"]), dp_mode=dp_mode,
                             persistent_workgroups_in_force=_ops.get_persistent_wgs(),
                             persistent_workgroups_note="of 256 CUs: forward / data-gradient / 5x5 launches; 224 by itself when world > 1 and the "
                                                        "bucketed all-reduce overlaps the backward pass (UGN_PERSISTENT_WGS overrides)",
                             gradient_bytes=grad_bytes, collectives_ms_per_step=coll_ms,
                             collectives_note="milliseconds per step per collective kind, event pairs on the issuing stream of every rank; "
                                              "max / min over the ranks")
        fpc = flop_per_clip(kinds)
        exec_factor = EXEC_FACTOR[args.dtype]
        exec_peak = PEAK_F32_MFMA if args.dtype == "f32" else PEAK_BF16_MFMA
        out = dict(metric="clips/sec (%s, L=25, 60x60) fwd+bwd+Adam" % ("3-mod" if len(kinds) == 3 else "%d-mod" % len(kinds)),
                   value=round(value, 2), unit="clips/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
                   higher_is_better=True, scaling=args.scaling, vs_baseline=None,
                   dtype=DTYPE_TEXT[args.dtype],
                   data="synthetic",
                   config=dict(workload=wl["text"] % (b_gpu, n_ids, max(1, b_gpu // n_ids) if args.scaling == "weak" else wl["ids_per"]),
                               clips_per_gpu=b_gpu, global_batch=world * b_gpu, parallelism="dp%d" % world, dp_mode=dp_mode,
                               masked_pairs_skipped=bool(args.skip_masked), distributed=dist_info,
                               launch_schedule="serial (one stream)" if args.serial else
                               ("one launch per layer for all modalities (frame-level layer + set-level twin as jobs of one "
                                "launch): forward = one chain on the main stream; weight gradients on a second stream beside the "
                                "data gradients; head forward beside the triplet kernel; filter repack on the second stream beside "
                                "the next step's 5x5 layer") + ("; replayed as two captured hipGraphs (--graph)" if args.graph else "")),
                   whole_step_tflops=round(value * fpc / 1e12, 2),
                   whole_step_frac_of_matrix_peak=round(value * fpc * exec_factor / world / exec_peak, 4),
                   loss=round(losses["loss"], 5), roofline=roof)
        if args.dtype in ("f32x3", "f32"):     # the fp32-tensor arithmetics: algorithmic FLOPs of the whole step against the fp32-MFMA peak
            out["whole_step_frac_of_f32_mfma_peak"] = round(value * fpc / world / PEAK_F32_MFMA, 4)
        out["whole_step_note"] = ("whole_step_tflops prices the step with SURVEY 8(d)'s algorithmic FLOPs per clip (71.3 G for 3 "
                                  "modalities); the fraction beside it counts the FLOPs the matrix pipe executes for them (f32x3: 6x "
                                  "against the dense bf16 peak of 2.5 PFLOP/s; Winograd f32: 16/36 against the fp32-MFMA peak; f16x2: 3x "
                                  "against the dense f16 peak); whole_step_frac_of_f32_mfma_peak = the ALGORITHMIC FLOPs of the whole step, "
                                  "head, losses, pooling and Adam included in the time, against 157.3 TFLOP/s: above 1 = faster than any kernel "
                                  "issuing fp32 matrix instructions could run the convolutions alone")
        if skip_rate is not None:
            out["value_skip_masked"] = round(skip_rate, 2)   # 29 of the 72 (clip, modality) pairs of the C3 batch are masked
        keep_roof = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_us", "launches_per_step", "share_of_step",
                     "algorithmic_tflops", "mfma_frac", "hbm_frac", "rocprof_kernel", "images_per_launch", "serial_step_us",
                     "algorithmic_frac_of_f32_mfma_peak", "stalled_launches")
        if all_present is not None:
            out["value_all_present"] = round(world * b_gpu * args.steps / all_present["dt"], 2)
            out["ms_per_step_all_present"] = round(all_present["dt"] / args.steps * 1e3, 3)
            out["loss_all_present"] = round(all_present["loss"], 5)
            out["all_present_note"] = ("the same shapes, arithmetic, steps and warm-up with every modality present in every clip (all use flags 1, "
                                       "no constant 1e-9 placeholder tensors): the dense rate on data with no constant frames")
            if all_present["roof"] is not None:
                out["roofline_all_present"] = {k: all_present["roof"].get(k) for k in keep_roof if all_present["roof"].get(k) is not None}
        for o in others:
            tag, prec = o["tag"], o["prec"]
            out["value_" + tag] = round(world * b_gpu * args.steps / o["dt"], 2)
            out["ms_per_step_" + tag] = round(o["dt"] / args.steps * 1e3, 3)
            out["dtype_" + tag] = DTYPE_TEXT[prec] + " (GaitCore(conv_precision=%r)); same batch, steps, warm-up and barriers as `value`" % prec
            out["loss_" + tag] = round(o["loss"], 5)
            out["whole_step_frac_of_matrix_peak_" + tag] = round(out["value_" + tag] * fpc * EXEC_FACTOR[prec] / world /
                                                                 (PEAK_F32_MFMA if prec == "f32" else PEAK_BF16_MFMA), 4)
            if o["roof"] is not None:
                out["roofline_" + tag] = {k: o["roof"].get(k) for k in keep_roof if o["roof"].get(k) is not None}
        if not args.no_cpu_baseline and world == 1:
            big = b_gpu > 40      # (the generator-expanded batches: time the workload's own batch on the CPU)
            out["cpu_baseline"] = cpu_baseline(kinds, ncls, wl["clips"] if big else b_gpu, wl["clips"] // wl["ids_per"] if big else n_ids,
                                               multimodal=multimodal)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = build_parser().parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)     # children only; this process never touches the GPU
        return
    run(args)


if __name__ == "__main__":
    main()
