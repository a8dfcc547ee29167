#!/usr/bin/env python3
"""In-kernel clock stamps of the f16-pipe convolution kernel (diagnostic build: python -m ugaitnet_amd.build --variant stamp
-DUGN_MM_STAMP; UGN_LIB=.../libugaitnet_hip_stamp.so python tools/stamp_mm.py [--layer a6] [--op fwd]).

Reports the clock the kernel ran at (s_memtime / s_memrealtime per workgroup) and, for a few workgroups, how long the
multiplying wave 0 and the loader waves 8 / 9 waited at each stage barrier.
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
PER_WAVE = 4 + 4 * 100


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
    buf = torch.zeros(256 * 12 * PER_WAVE, dtype=torch.int64, device=dev)
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
    st = buf.cpu().numpy().reshape(256, 12, PER_WAVE)
    t0, r0, t1, r1 = st[:, 0, 0], st[:, 0, 1], st[:, 0, 2], st[:, 0, 3]
    ok = t1 > 0
    cyc, real = (t1 - t0)[ok].astype(np.float64), (r1 - r0)[ok].astype(np.float64)
    print("%s %s: workgroups %d; kernel time per workgroup median %.1f us (max %.1f); in-kernel clock median %.3f GHz"
          % (layer, op, ok.sum(), np.median(real) / 100.0, real.max() / 100.0, np.median(cyc / real) * 0.1))
    for wg in (0, 100):
        print("workgroup %d, per stage: period | multiplier wave 0: barrier wait, body | filter loader 8: vmcnt wait, barrier wait, "
              "issue | tile loader 10: vmcnt wait, barrier wait, issue   (cycles)" % wg)
        q = lambda w: st[wg, w, 4:].reshape(-1, 4).astype(np.float64)
        m, fl, tl = q(0), q(8), q(10)
        n = int((m[:, 2] > 0).sum())
        rows = []
        for s in range(1, min(n, 100)):
            rows.append((m[s, 2] - m[s - 1, 2], m[s, 2] - m[s, 1], m[s, 3] - m[s, 2],
                         fl[s, 1] - fl[s, 0], fl[s, 2] - fl[s, 1], fl[s, 3] - fl[s, 2],
                         tl[s, 1] - tl[s, 0], tl[s, 2] - tl[s, 1], tl[s, 3] - tl[s, 2]))
        rows = np.array(rows)
        for s in range(min(len(rows), 45)):
            print("  stage %3d: %6d | %6d %6d | %6d %6d %6d | %6d %6d %6d" % (s + 1, *rows[s]))
        print("  mean over %d stages: %s" % (len(rows), " ".join("%.0f" % v for v in rows.mean(axis=0))))
        allw = st[wg, :, 4:].reshape(12, -1, 4).astype(np.float64)
        bw = allw[:, 1:n, 2] - allw[:, 1:n, 1]          # barrier wait of every wave
        body = allw[:, 1:n, 3] - allw[:, 1:n, 2]
        arrive = allw[:, 1:n, 1] - allw[0:1, 1:n, 2] + bw[0:1]     # arrival relative to the release seen by wave 0
        print("  mean barrier wait per wave: " + " ".join("%.0f" % v for v in bw.mean(axis=1)))
        print("  mean body per wave:         " + " ".join("%.0f" % v for v in body.mean(axis=1)))
        last = bw.argmin(axis=0)
        print("  last arriver histogram:     " + " ".join("%d" % (last == w).sum() for w in range(12)))


if __name__ == "__main__":
    main()
