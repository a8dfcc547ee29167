#!/usr/bin/env python3
"""Winograd fp32-MFMA kernels against the x3 kernels (fp32 tensors, three-way bf16 split on the bf16 pipe) on the merged launches
of the C3 step (3 modalities x 600 frames + 3 x 24 set-level maps per launch), one process, HIP-event medians.

    python tools/bench_x3.py [--frames 600] [--reps 20] [--ops fwd,dgrad,wgrad] [--layers a2,a3,a4,a5,a6] [--only-x3] [--cold]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ugaitnet_amd import ops, x3

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}
PEAK = 2516.8e12


def main():
    argv = sys.argv[1:]
    opt = lambda k, d: argv[argv.index(k) + 1] if k in argv else d
    frames, reps = int(opt("--frames", 600)), int(opt("--reps", 20))
    kinds = opt("--ops", "fwd,dgrad,wgrad").split(",")
    layers = opt("--layers", "a2,a3,a4,a5,a6").split(",")
    nmod = int(opt("--mods", 3))
    only = "--only-x3" in argv
    dev = torch.device("cuda")
    res = {}
    cold = "--cold" in argv
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None

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
        ws = [torch.randn(3, 3, cin, cout, device=dev) * 0.05 for _ in ns]
        dzs = [torch.randn(n, ho, ho, cout, device=dev) * 1e-4 for n in ns]
        idxs = [torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
        flops = 2.0 * 9 * cin * cout * hw * hw * sum(ns)
        fns = {}
        if "fwd" in kinds:
            outs = [torch.empty(n, ho, ho, cout, device=dev) for n in ns]
            oidx = [torch.empty(n, ho, ho, cout, device=dev, dtype=torch.uint8) for n in ns] if pool else None
            if not only:
                ufs = [ops.wino_pack(w, False) for w in ws]
                fns["fwd wino"] = lambda: ops.conv3x3_fwd_wino_multi(xs, ufs, cout, pool, outs, oidx)
            pk = [x3.pack(w, False) for w in ws]
            fns["fwd x3"] = lambda: x3.conv3x3_fwd_multi(xs, pk, cout, pool, outs, oidx)
        if "dgrad" in kinds:
            douts = [torch.empty(n, hw, hw, cin, device=dev) for n in ns]
            use_act = layer in ("a4", "a6")
            acts = xs if use_act else None
            if not only:
                uds = [ops.wino_pack(w, True, pooled_dz=pool) for w in ws]
                fns["dgrad wino"] = lambda: ops.conv3x3_dgrad_wino_multi(dzs, uds, hw, cin, cout, douts, dz_idxs=idxs, acts=acts)
            pkd = [x3.pack(w, True) for w in ws]
            fns["dgrad x3"] = lambda: x3.conv3x3_dgrad_multi(dzs, pkd, hw, cin, cout, douts, dz_idxs=idxs, acts=acts)
        if "wgrad" in kinds:
            dws = [torch.empty(3, 3, cin, cout, device=dev) for _ in ns]
            if not only:
                fns["wgrad wino"] = lambda: ops.conv3x3_wgrad_wino_multi(xs, dzs, cout, dws, dz_idxs=idxs)
            fns["wgrad x3"] = lambda: x3.conv3x3_wgrad_multi(xs, dzs, cout, dws, dz_idxs=idxs)
        for name, fn in fns.items():
            us = timeit(fn)
            row = dict(us=round(us, 1), algorithmic_tflops=round(flops / us / 1e6, 1))
            if name.endswith("x3"):
                row["mfma_frac"] = round(flops * 6 / (us * 1e-6) / PEAK, 3)
            res["%s %s" % (layer, name)] = row
            print("%-16s %8.1f us  %7.1f algorithmic TFLOP/s %s" % (layer + " " + name, us, flops / us / 1e6,
                                                                    ("  %.3f of the bf16 peak executed" % row["mfma_frac"]) if "mfma_frac" in row else ""), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
