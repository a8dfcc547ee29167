#!/usr/bin/env python3
"""Winograd fp32 kernels against the f16-matrix-pipe kernels on H2 tensors, on the merged launches of the C3 step
(3 modalities x 600 frames + 3 x 24 set-level maps per launch), one process, interleaved rounds, HIP-event medians.

    python tools/bench_mm.py [--frames 600] [--reps 20] [--rounds 3] [--ops fwd,dgrad,wgrad] [--layers a2,a3,a4,a5,a6]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ugaitnet_amd import h2, ops

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}


def main():
    argv = sys.argv[1:]
    opt = lambda k, d: argv[argv.index(k) + 1] if k in argv else d
    frames, reps, rounds = int(opt("--frames", 600)), int(opt("--reps", 20)), int(opt("--rounds", 3))
    kinds = opt("--ops", "fwd,dgrad,wgrad").split(",")
    layers = opt("--layers", "a2,a3,a4,a5,a6").split(",")
    nmod = int(opt("--mods", 3))
    dev = torch.device("cuda")
    res = {}

    cold = "--cold" in argv      # evict the Infinity Cache (256 MB) before every timed launch: in the training step a kernel's
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None   # inputs come from HBM, not from a replay

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            if cold:
                junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ts]) * 1e3)

    for layer in layers:
        hw, cin, cout, pool = CFGS[layer]
        ns = [frames] * nmod + ([24] * nmod if layer != "a2" else [])
        ho = hw // 2 if pool else hw
        xs = [torch.randn(n, hw, hw, cin, device=dev) for n in ns]
        if "--zeros" in argv:       # all-zero operands: same instructions and cycles, less switching power -- a kernel that gets faster on
            for x in xs:            # zeros was held back by the clock the chip sustains under load, not by its instruction schedule
                x.zero_()
        if "--masked" in argv:      # 40 % of the images carry the tiny activations of a masked modality (input = constant 1e-9)
            for x in xs:
                x[torch.rand(x.shape[0], device=dev) < 0.4] = 1e-9
        ws = [torch.randn(3, 3, cin, cout, device=dev) * (0.0 if "--zeros" in argv else 0.05) for _ in ns]
        dzs = [torch.randn(n, ho, ho, cout, device=dev) * (0.0 if "--zeros" in argv else 1e-4) for n in ns]
        idxs = [torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
        flops = 2.0 * 9 * cin * cout * hw * hw * sum(ns)
        fns = {}
        if "fwd" in kinds:
            ufs = [ops.wino_pack(w, False) for w in ws]
            outs = [torch.empty(n, ho, ho, cout, device=dev) for n in ns]
            oidx = [torch.empty(n, ho, ho, cout, device=dev, dtype=torch.uint8) for n in ns] if pool else None
            if "--only-mm" not in argv:
                fns["fwd wino"] = lambda: ops.conv3x3_fwd_wino_multi(xs, ufs, cout, pool, outs, oidx)
            hx = [h2.encode(x) for x in xs]
            pk = [h2.mm_pack(w, False) for w in ws]
            ho_ = [h2.H2Tensor.empty((n, ho, ho, cout), dev) for n in ns]
            fns["fwd mm"] = lambda: h2.conv3x3_fwd_mm_multi(hx, [p[0] for p in pk], [p[1] for p in pk], cout, pool, ho_, oidx)
        if "dgrad" in kinds:
            uds = [ops.wino_pack(w, True, pooled_dz=pool) for w in ws]
            douts = [torch.empty(n, hw, hw, cin, device=dev) for n in ns]
            use_act = layer in ("a4", "a6")
            acts = xs if use_act else None
            if "--only-mm" not in argv:
                fns["dgrad wino"] = lambda: ops.conv3x3_dgrad_wino_multi(dzs, uds, hw, cin, cout, douts, dz_idxs=idxs, acts=acts)
            hdz = [h2.encode(d) for d in dzs]
            pkd = [h2.mm_pack(w, True) for w in ws]
            hdo = [h2.H2Tensor.empty((n, hw, hw, cin), dev) for n in ns]
            hact = [h2.encode(x) for x in xs] if use_act else None
            fns["dgrad mm"] = lambda: h2.conv3x3_dgrad_mm_multi(hdz, [p[0] for p in pkd], [p[1] for p in pkd], hw, cin, cout, hdo,
                                                                                 dz_idxs=idxs, acts=hact)
        if "wgrad" in kinds:
            dws = [torch.empty(3, 3, cin, cout, device=dev) for _ in ns]
            if "--only-mm" not in argv:
                fns["wgrad wino"] = lambda: ops.conv3x3_wgrad_wino_multi(xs, dzs, cout, dws, dz_idxs=idxs)
            if hasattr(h2, "conv3x3_wgrad_mm_multi"):
                hx2 = [h2.encode(x) for x in xs]
                hdz2 = [h2.encode(d) for d in dzs]
                dws2 = [torch.empty(3, 3, cin, cout, device=dev) for _ in ns]
                fns["wgrad mm"] = lambda: h2.conv3x3_wgrad_mm_multi(hx2, hdz2, cout, dws2, dz_idxs=idxs)
        times = {k: [] for k in fns}
        for _ in range(rounds):
            for k, fn in fns.items():
                times[k].append(timeit(fn))
        for k, v in times.items():
            t = float(np.median(v))
            res["%s %s" % (layer, k)] = t
            print("%-4s %-12s %8.1f us   %6.1f algorithmic TFLOP/s   (rounds: %s)" % (layer, k, t, flops / t * 1e-6, " ".join("%.0f" % x for x in v)), flush=True)
    print("BENCH_MM " + json.dumps(res))


# (the output metas gather their maximum by atomicMax and are NOT zeroed between the timed launches: the data does not
#  change, so the maximum does not either; the engine zeroes all of them with one memset per step)
if __name__ == "__main__":
    main()
