"""Micro-benchmark of the 5x5 first-layer kernels (600 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops

dev = torch.device("cuda")
N = 600


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rep in range(2):
    for cin in (1, 2):
        x = torch.randn(N, 60, 60, cin, device=dev)
        w = torch.randn(5, 5, cin, 32, device=dev) * 0.1
        dz = torch.randn(N, 64, 64, 32, device=dev)
        a1 = torch.empty(N, 64, 64, 32, device=dev)
        t_f = timeit(lambda: ops.conv5x5_in_fwd(x, w, a1))
        t_w = timeit(lambda: ops.conv5x5_in_wgrad(x, dz))
        print("cin=%d fwd %6.1f us (%4.2f TB/s written) | wgrad %6.1f us (%4.2f TB/s read)" % (
            cin, t_f, a1.numel() * 4 / t_f / 1e6, t_w, dz.numel() * 4 / t_w / 1e6), flush=True)
