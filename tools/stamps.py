#!/usr/bin/env python3
"""Where a group of a Winograd forward / data-gradient launch spends its cycles (diagnostic build only).

    python -m ugaitnet_amd.build --variant stamps -DUGN_STAMPS=1
    UGN_LIB=ugaitnet_amd/libugaitnet_hip_stamps.so python tools/stamps.py a6_fwd [frames]

Workgroup 0 stamps s_memtime before the wait + barrier that opens every channel group and right after it (lane 0 of each wave).
Printed per wave: cycles inside the groups (barrier exit -> next barrier entry), cycles in wait + barrier, and the same per item."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ugaitnet_amd import _lib, ops
from tools.ab_ops import CFGS


def main():
    layer, kind = sys.argv[1].split("_")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    hw, cin, cout, pool = CFGS[layer]
    dev = torch.device("cuda")
    x = torch.randn(n, hw, hw, cin, device=dev)
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    ho = hw // 2 if pool else hw
    if kind == "fwd":
        uf = ops.wino_pack(w, False)
        run = lambda: ops.conv3x3_fwd_wino(x, uf, cout, pool)
    else:
        ud = ops.wino_pack(w, True, pooled_dz=pool)
        dz = torch.randn(n, ho, ho, cout, device=dev)
        idx = torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) if pool else None
        act = torch.randn(n, hw, hw, cin, device=dev) if layer in ("a4", "a6") else None
        run = lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_ulonglong * (8 * 2048))()
    lib.ugn_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.ugn_debug_stamps(buf, 8 * 2048) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 2048).astype(np.int64)
    for wv in range(8):
        t = a[wv]
        k = int(np.nonzero(t)[0].max()) + 1 if t.any() else 0
        t = t[:k - (k % 2)]
        before, after = t[0::2], t[1::2]            # before wait+barrier, after barrier
        wait = after - before
        inside = before[1:] - after[:-1]
        print("wave %d: %4d groups  inside median %6.0f mean %6.0f  |  wait+barrier median %5.0f mean %6.0f  | total/group %6.0f"
              % (wv, len(wait), np.median(inside), inside.mean(), np.median(wait), wait.mean(), (t[-1] - t[0]) / max(1, len(wait) - 1)))
    t = a[0]
    k = int(np.nonzero(t)[0].max()) + 1
    t = t[:k - (k % 2)]
    inside = (t[0::2][1:] - t[1::2][:-1])
    print("wave 0 inside-group cycles, first 40 groups:", inside[:40].tolist())
    print("wave 0 wait+barrier cycles, first 40 groups:", (t[1::2] - t[0::2])[:40].tolist())


if __name__ == "__main__":
    main()
