#!/usr/bin/env python3
"""Where a workgroup of an x3 forward / data-gradient launch spends its cycles (diagnostic build only).

    python -m ugaitnet_amd.build --variant x3stamp -DUGN_X3_STAMP=1
    UGN_LIB=ugaitnet_amd/libugaitnet_hip_x3stamp.so python tools/stamp_x3.py a2_fwd [a2_dgrad ...] [--frames 600]

Wave 0 of workgroups 0..31 stamps s_memtime per item: 0 item start, 1 behind the first barrier (tile complete), 2 behind the last
chunk's matrix loop, 3 behind the issue of the next tile's loads (4-wave form), 4 behind the epilogue, 5 end of item (4-wave form:
next tile split and written); 6 = s_memrealtime at the start (100 MHz) for the clock."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ugaitnet_amd import _lib, x3
from tools.ab_ops import CFGS

NWG, NIT = 32, 48


def main():
    argv = sys.argv[1:]
    frames = int(argv[argv.index("--frames") + 1]) if "--frames" in argv else 600
    ops_ = [a for a in argv if "_" in a]
    dev = torch.device("cuda")
    lib = _lib.load()
    lib.ugn_x3_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    for op in ops_:
        layer, kind = op.split("_")
        hw, cin, cout, pool = CFGS[layer]
        ns = [frames] * 3
        ho = hw // 2 if pool else hw
        xs = [torch.randn(n, hw, hw, cin, device=dev) for n in ns]
        ws = [torch.randn(3, 3, cin, cout, device=dev) * 0.05 for _ in ns]
        if kind == "fwd":
            outs = [torch.empty(n, ho, ho, cout, device=dev) for n in ns]
            oidx = [torch.empty(n, ho, ho, cout, device=dev, dtype=torch.uint8) for n in ns] if pool else None
            pk = [x3.pack(w, False) for w in ws]
            run = lambda: x3.conv3x3_fwd_multi(xs, pk, cout, pool, outs, oidx)
        else:
            dzs = [torch.randn(n, ho, ho, cout, device=dev) * 1e-4 for n in ns]
            idxs = [torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
            douts = [torch.empty(n, hw, hw, cin, device=dev) for n in ns]
            acts = xs if layer in ("a4", "a6") else None
            pkd = [x3.pack(w, True) for w in ws]
            run = lambda: x3.conv3x3_dgrad_multi(dzs, pkd, hw, cin, cout, douts, dz_idxs=idxs, acts=acts)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        scratch = (C.c_ulonglong * (NWG * NIT * 8))()
        assert lib.ugn_x3_debug_stamps(scratch, NWG * NIT * 8) == 0          # (reading clears the buffer)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * (NWG * NIT * 8))()
        assert lib.ugn_x3_debug_stamps(buf, NWG * NIT * 8) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(NWG, NIT, 8).astype(np.int64)
        nvalid = int((a[:, :, 5] > 0).all(axis=0).sum())          # items every stamped workgroup has run
        a = a[:, min(3, nvalid // 4):max(2, nvalid - 1)]          # (skip the pipeline fill and the last item)
        clk = (a[:, -1, 0] - a[:, 0, 0]) / np.maximum(1, (a[:, -1, 6] - a[:, 0, 6])) * 100e6
        d = lambda i, j: np.median(a[:, :, j] - a[:, :, i])
        per_item = np.median(a[:, 1:, 0] - a[:, :-1, 0])
        print("%s: launch %.1f us, clock %.2f GHz (median over workgroups), cycles per item %d = %.2f us" % (op, e0.elapsed_time(e1) * 1e3, np.median(clk) / 1e9, per_item, per_item / np.median(clk) * 1e6))
        print("   wait for the tile (0->1) %d | matrix loop(s) (1->2) %d | issue of the next loads (2->3) %d | epilogue (3->4) %d | split + LDS writes behind it (4->5) %d cycles"
              % (d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5)))
        print("   per workgroup, cycles per item:", np.median(a[:, 1:, 0] - a[:, :-1, 0], axis=1)[:16].astype(int).tolist())


if __name__ == "__main__":
    main()
