#!/usr/bin/env python3
"""In-kernel clock stamps of the f16-pipe weight-gradient kernel (diagnostic build: python -m ugaitnet_amd.build --variant wgstamp
-DUGN_WG_STAMP; UGN_LIB=.../libugaitnet_hip_wgstamp.so python tools/stamp_wgrad.py [--layer a2]).  Per strip and wave: wait for the
tiles, barrier, pooled scatter (+ barrier), issue of the next strip's LDS-DMA, transposed reads + MFMAs."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ugaitnet_amd import _lib, h2

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}
PER_WAVE = 4 + 6 * 60


def main():
    argv = sys.argv[1:]
    opt = lambda k, d: argv[argv.index(k) + 1] if k in argv else d
    layer, frames = opt("--layer", "a2"), int(opt("--frames", 600))
    dev = torch.device("cuda")
    hw, cin, cout, pool = CFGS[layer]
    ns = [frames] * 3 + ([24] * 3 if layer != "a2" else [])
    ho = hw // 2 if pool else hw
    xs = [h2.encode(torch.randn(n, hw, hw, cin, device=dev)) for n in ns]
    dzs = [h2.encode(torch.randn(n, ho, ho, cout, device=dev) * 1e-4) for n in ns]
    idxs = [torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
    dws = [torch.empty(3, 3, cin, cout, device=dev) for _ in ns]
    fn = lambda: h2.conv3x3_wgrad_mm_multi(xs, dzs, cout, dws, dz_idxs=idxs)
    lib = _lib.load()
    lib.ugn_wg_debug_stamps.argtypes = [C.c_void_p]
    lib.ugn_wg_debug_stamps.restype = C.c_int
    buf = torch.zeros(256 * 8 * PER_WAVE, dtype=torch.int64, device=dev)
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    assert lib.ugn_wg_debug_stamps(C.c_void_p(buf.data_ptr())) == 0
    fn()
    torch.cuda.synchronize()
    lib.ugn_wg_debug_stamps(C.c_void_p(0))
    st = buf.cpu().numpy().reshape(256, 8, PER_WAVE)
    t0, r0, t1, r1 = st[:, 0, 0], st[:, 0, 1], st[:, 0, 2], st[:, 0, 3]
    ok = t1 > 0
    cyc, real = (t1 - t0)[ok].astype(np.float64), (r1 - r0)[ok].astype(np.float64)
    print("%s wgrad: workgroups %d; kernel time per workgroup median %.1f us (max %.1f); in-kernel clock median %.3f GHz"
          % (layer, ok.sum(), np.median(real) / 100.0, real.max() / 100.0, np.median(cyc / real) * 0.1))
    names = ["tile wait", "barrier", "scatter+barrier", "DMA issue", "reads+MFMA", "loop tail"]
    for wg in (0, 100):
        q = st[wg, :, 4:].reshape(8, -1, 6).astype(np.float64)
        n = int((q[0, :, 5] > 0).sum()) - 1
        seg = np.stack([q[:, 1:n, 1] - q[:, 1:n, 0], q[:, 1:n, 2] - q[:, 1:n, 1], q[:, 1:n, 3] - q[:, 1:n, 2], q[:, 1:n, 4] - q[:, 1:n, 3],
                        q[:, 1:n, 5] - q[:, 1:n, 4], q[:, 2:n + 1, 0] - q[:, 1:n, 5]], axis=-1)     # [wave][strip][segment]
        period = (q[0, 2:n + 1, 2] - q[0, 1:n, 2]).mean()
        print("workgroup %d: %d strips, period %.0f cycles; mean cycles per segment and wave:" % (wg, n, period))
        for k, nm in enumerate(names):
            print("  %-16s %s" % (nm, " ".join("%6.0f" % v for v in seg[:, :, k].mean(axis=1))))


if __name__ == "__main__":
    main()
