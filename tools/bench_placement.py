#!/usr/bin/env python3
"""How the placement of a kernel's output tensor relative to its input tensor changes its duration (HBM channel / bank
aliasing): one arena, input at offset 0, output at offset D, forward 32 -> 64 @32x32 on 600 frames (the a3 launch of one
modality).   python tools/bench_placement.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ugaitnet_amd import h2

dev = torch.device("cuda")
MiB = 1 << 20
n, hw, cin, cout = 600, 32, 32, 64
in_bytes, out_bytes = n * hw * hw * cin * 4, n * hw * hw * cout * 4
arena = torch.empty(2048 * MiB, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
print("arena at 0x%x (mod 2 MiB = %d); input %d MiB, output %d MiB" % (base, base % (2 * MiB), in_bytes // MiB, out_bytes // MiB))


def view(off, shape):
    nbytes = int(np.prod(shape)) * 2
    return arena[off:off + nbytes].view(torch.int16).view(shape)


x = torch.randn(n, hw, hw, cin, device=dev)
w = torch.randn(3, 3, cin, cout, device=dev) * 0.05
pk, wm = h2.mm_pack(w, False)
xin = h2.H2Tensor(view(0, (n, hw, hw, 2, cin)), torch.zeros(2, dtype=torch.int32, device=dev))
h2.encode(x, out=xin)


def timeit(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ts]) * 1e3)


KiB = 1024
deltas = [76 * MiB, 75 * MiB, 80 * MiB, 96 * MiB, 128 * MiB, 150 * MiB, 192 * MiB, 256 * MiB, 300 * MiB, 512 * MiB, 1024 * MiB,
          76 * MiB + 4 * KiB, 76 * MiB + 64 * KiB, 76 * MiB + 256 * KiB, 76 * MiB + 1 * MiB, 128 * MiB + 4 * KiB, 128 * MiB + 64 * KiB,
          128 * MiB + 512 * KiB, 256 * MiB + 64 * KiB, 256 * MiB + 1 * MiB, 77 * MiB, 78 * MiB, 79 * MiB, 84 * MiB, 88 * MiB, 100 * MiB, 112 * MiB]
for d in deltas:
    out = h2.H2Tensor(view(d, (n, hw, hw, 2, cout)), torch.zeros(2, dtype=torch.int32, device=dev))
    t = timeit(lambda: h2.conv3x3_fwd_mm_multi([xin], [pk], [wm], cout, False, [out]))
    print("output at +%8.3f MiB (+%d B): %7.1f us" % (d / MiB, d, t), flush=True)
# separate allocations, as the engine makes them
outs = [h2.H2Tensor.empty((n, hw, hw, cout), dev) for _ in range(3)]
for o in outs:
    t = timeit(lambda: h2.conv3x3_fwd_mm_multi([xin], [pk], [wm], cout, False, [o]))
    print("separate allocation at 0x%x (delta %+.3f MiB): %7.1f us" % (o.data.data_ptr(), (o.data.data_ptr() - base) / MiB, t))
