#!/usr/bin/env python3
"""Error against the fp64 oracle of the three fp32-tensor implementations of every 3x3 layer shape, on the same inputs:
x3 (three-way bf16 split on the bf16 pipe), the direct fp32-MFMA kernels (UGN_WINO=0 path) and the Winograd fp32-MFMA kernels.
Prints max |err| / max |ref| and rms err / rms ref per (layer, operator).

    python tools/x3_accuracy.py [--n 6]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import ugaitnet_oracle as O
from ugaitnet_amd import ops, x3

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}


def errs(got, ref):
    g = got.detach().cpu().numpy().astype(np.float64)
    d = g - ref
    return float(np.abs(d).max() / np.abs(ref).max()), float(np.sqrt((d * d).mean()) / np.sqrt((ref * ref).mean()))


def main():
    argv = sys.argv[1:]
    n = int(argv[argv.index("--n") + 1]) if "--n" in argv else 6
    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(11)
    res = {}
    for layer, (hw, cin, cout, pool) in CFGS.items():
        nn = n if hw <= 32 else max(2, n // 3)
        x = rng.uniform(-1, 1, (nn, hw, hw, cin)).astype(np.float32)
        w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        hz = hw // 2 if pool else hw
        dzf = (rng.standard_normal((nn, hz, hz, cout)) * 1e-3).astype(np.float32)
        pidx = rng.integers(0, 4, size=dzf.shape).astype(np.uint8) if pool else None
        xt, wt, dzt = T(x), T(w), T(dzf)
        it = T(pidx) if pool else None
        x64, w64 = x.astype(np.float64), w.astype(np.float64)
        pre = O.conv2d_same(x64, w64)
        act = O.leaky(pre)
        fref = O.maxpool2x2(act)[0] if pool else act
        dz64 = O.maxpool2x2_bwd(pidx, dzf.astype(np.float64)) if pool else dzf.astype(np.float64)
        dw_ref, dx_ref = O.conv2d_same_bwd(x64, w64, dz64)
        ho = hz
        rows = {}
        # forward
        o = torch.empty((nn, ho, ho, cout), device=dev)
        oi = torch.empty((nn, ho, ho, cout), dtype=torch.uint8, device=dev) if pool else None
        x3.conv3x3_fwd_multi([xt], [x3.pack(wt, False)], cout, pool, [o], [oi] if pool else None)
        rows["fwd x3"] = errs(o, fref)
        x3.conv3x3_fwd_multi([xt], [x3.pack(wt, False)], cout, pool, [o], [oi] if pool else None, products=9)
        rows["fwd x3 (9 products)"] = errs(o, fref)
        r = ops.conv3x3_fwd(xt, ops.pack3x3(wt), pool)
        rows["fwd direct"] = errs(r[0] if pool else r, fref)
        r = ops.conv3x3_fwd_wino(xt, ops.wino_pack(wt, False), cout, pool)
        rows["fwd wino"] = errs(r[0] if pool else r, fref)
        # data gradient
        d = torch.empty((nn, hw, hw, cin), device=dev)
        x3.conv3x3_dgrad_multi([dzt], [x3.pack(wt, True)], hw, cin, cout, [d], dz_idxs=[it] if pool else None)
        rows["dgrad x3"] = errs(d, dx_ref)
        x3.conv3x3_dgrad_multi([dzt], [x3.pack(wt, True)], hw, cin, cout, [d], dz_idxs=[it] if pool else None, products=9)
        rows["dgrad x3 (9 products)"] = errs(d, dx_ref)
        rows["dgrad direct"] = errs(ops.conv3x3_dgrad(dzt, wt, hw, dz_idx=it), dx_ref)
        rows["dgrad wino"] = errs(ops.conv3x3_dgrad_wino(dzt, ops.wino_pack(wt, True, pooled_dz=pool), hw, cin, cout, dz_idx=it), dx_ref)
        # weight gradient
        g = torch.empty((3, 3, cin, cout), device=dev)
        x3.conv3x3_wgrad_multi([xt], [dzt], cout, [g], dz_idxs=[it] if pool else None)
        rows["wgrad x3"] = errs(g, dw_ref)
        x3.conv3x3_wgrad_multi([xt], [dzt], cout, [g], dz_idxs=[it] if pool else None, products=9)
        rows["wgrad x3 (9 products)"] = errs(g, dw_ref)
        rows["wgrad direct"] = errs(ops.conv3x3_wgrad(xt, dzt, cout, dz_idx=it), dw_ref)
        rows["wgrad wino"] = errs(ops.conv3x3_wgrad_wino(xt, dzt, cout, dz_idx=it), dw_ref)
        for k, (mx, rms) in rows.items():
            print("%s %-22s max %.3e  rms %.3e" % (layer, k, mx, rms), flush=True)
            res["%s %s" % (layer, k)] = dict(max=mx, rms=rms)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
