#!/usr/bin/env python3
"""In-kernel clock stamps of the f16-pipe convolution kernel (diagnostic build: python -m ugaitnet_amd.build --variant stamp
-DUGN_MM_STAMP; UGN_LIB=.../libugaitnet_hip_stamp.so python tools/stamp_mm.py [--layer a6] [--op fwd]).

Reports the clock the kernel ran at (s_memtime / s_memrealtime per workgroup) and, for two workgroups, the mean cycles every
wave spends in the phases of a stage (wait for the LDS-DMA, barrier, issue of the next DMA pieces, LDS reads + MFMAs) and in the
epilogue of an item.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ugaitnet_amd import _lib, h2

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}
NST, NIT = 60, 20
PER_WAVE = 4 + 5 * NST + 4 * NIT


def main():
    argv = sys.argv[1:]
    opt = lambda k, d: argv[argv.index(k) + 1] if k in argv else d
    layer, op, frames = opt("--layer", "a6"), opt("--op", "fwd"), int(opt("--frames", 600))
    dev = torch.device("cuda")
    hw, cin, cout, pool = CFGS[layer]
    ns = [frames] * 3 + ([24] * 3 if layer != "a2" else [])
    ho = hw // 2 if pool else hw
    xs = [torch.randn(n, hw, hw, cin, device=dev) for n in ns]
    ws = [torch.randn(3, 3, cin, cout, device=dev) * 0.05 for _ in ns]
    lib = _lib.load()
    lib.ugn_mm_debug_stamps.argtypes = [C.c_void_p]
    lib.ugn_mm_debug_stamps.restype = C.c_int
    buf = torch.zeros(256 * 8 * PER_WAVE, dtype=torch.int64, device=dev)
    if op == "fwd":
        hx = [h2.encode(x) for x in xs]
        pk = [h2.mm_pack(w, False) for w in ws]
        out = [h2.H2Tensor.empty((n, ho, ho, cout), dev) for n in ns]
        oidx = [torch.empty(n, ho, ho, cout, device=dev, dtype=torch.uint8) for n in ns] if pool else None
        fn = lambda: h2.conv3x3_fwd_mm_multi(hx, [p[0] for p in pk], [p[1] for p in pk], cout, pool, out, oidx)
    else:
        dzs = [torch.randn(n, ho, ho, cout, device=dev) * 1e-4 for n in ns]
        idxs = [torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
        hdz = [h2.encode(d) for d in dzs]
        pkd = [h2.mm_pack(w, True) for w in ws]
        hdo = [h2.H2Tensor.empty((n, hw, hw, cin), dev) for n in ns]
        hact = [h2.encode(x) for x in xs] if layer in ("a4", "a6") else None
        fn = lambda: h2.conv3x3_dgrad_mm_multi(hdz, [p[0] for p in pkd], [p[1] for p in pkd], hw, cin, cout, hdo, dz_idxs=idxs, acts=hact)
    for _ in range(200):          # (the clock settles under load)
        fn()
    torch.cuda.synchronize()
    assert lib.ugn_mm_debug_stamps(C.c_void_p(buf.data_ptr())) == 0
    fn()
    torch.cuda.synchronize()
    lib.ugn_mm_debug_stamps(C.c_void_p(0))
    st = buf.cpu().numpy().reshape(256, 8, PER_WAVE).astype(np.float64)
    t0, r0, t1, r1 = st[:, 0, 0], st[:, 0, 1], st[:, 0, 2], st[:, 0, 3]
    ok = t1 > 0
    cyc, real = (t1 - t0)[ok], (r1 - r0)[ok]
    print("%s %s: workgroups %d; kernel time per workgroup median %.1f us (max %.1f); in-kernel clock median %.3f GHz"
          % (layer, op, ok.sum(), np.median(real) / 100.0, real.max() / 100.0, np.median(cyc / real) * 0.1))
    names = ["DMA wait", "barrier", "DMA issue (+scatter)", "reads + MFMA", "to next stage"]
    for wg in (0, 100):
        q = st[wg, :, 4:4 + 5 * NST].reshape(8, NST, 5)
        it = st[wg, :, 4 + 5 * NST:].reshape(8, NIT, 4)
        n = int((q[0, :, 4] > 0).sum()) - 1
        seg = np.stack([q[:, 1:n, 1] - q[:, 1:n, 0], q[:, 1:n, 2] - q[:, 1:n, 1], q[:, 1:n, 3] - q[:, 1:n, 2], q[:, 1:n, 4] - q[:, 1:n, 3],
                        q[:, 2:n + 1, 0] - q[:, 1:n, 4]], axis=-1)
        period = (q[0, 2:n + 1, 2] - q[0, 1:n, 2]).mean()
        print("workgroup %d: %d stages, mean period %.0f cycles (item boundaries included); mean cycles per segment and wave:" % (wg, n, period))
        for k, nm in enumerate(names):
            print("  %-22s %s" % (nm, " ".join("%6.0f" % v for v in seg[:, :, k].mean(axis=1))))
        m = int((it[0, :, 1] > 0).sum())
        if m > 2:
            print("  %-22s %s" % ("epilogue", " ".join("%6.0f" % v for v in (it[:, 1:m, 1] - it[:, 1:m, 0]).mean(axis=1))))
            print("  %-22s %s" % ("item period", " ".join("%6.0f" % v for v in (it[:, 2:m, 0] - it[:, 1:m - 1, 0]).mean(axis=1))))
            if (it[:, 1:m, 3] > 0).any():      # (register-staged pooled scatter: wait for the loads + scatter, inside the taps)
                print("  %-22s %s" % ("scatter (in taps)", " ".join("%6.0f" % v for v in (it[:, 1:m, 3] - it[:, 1:m, 2]).mean(axis=1))))
                print("  %-22s %s" % ("scatter start - top", " ".join("%6.0f" % v for v in (it[:, 1:m, 2] - q[:, 1:m, 0]).mean(axis=1))))


if __name__ == "__main__":
    main()
