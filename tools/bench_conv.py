"""Micro-benchmark of the 3x3 conv kernels at the frame-level shapes of the C3 workload (600 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops

CFGS = [(64, 32, 32, True, "a2"), (32, 32, 64, False, "a3"), (32, 64, 64, True, "a4"), (16, 64, 128, False, "a5"), (16, 128, 128, False, "a6")]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda")


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for hw, cin, cout, pool, name in CFGS:
    x = torch.randn(N, hw, hw, cin, device=dev)
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    wp, uf, ud = ops.pack3x3(w), ops.wino_pack(w, False), ops.wino_pack(w, True, pooled_dz=pool)
    ho = hw // 2 if pool else hw
    dz = torch.randn(N, ho, ho, cout, device=dev)
    idx = torch.randint(0, 4, (N, ho, ho, cout), device=dev, dtype=torch.uint8) if pool else None
    act = torch.randn(N, hw, hw, cin, device=dev)
    gf = 2.0 * 9 * cin * cout * hw * hw * N / 1e9
    t_fd = timeit(lambda: ops.conv3x3_fwd(x, wp, pool))
    t_fw = timeit(lambda: ops.conv3x3_fwd_wino(x, uf, cout, pool))
    t_dd = timeit(lambda: ops.conv3x3_dgrad(dz, w, hw, dz_idx=idx, act=act))
    t_dw = timeit(lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act))
    t_wg = timeit(lambda: ops.conv3x3_wgrad(x, dz, cout, dz_idx=idx))
    t_ww = timeit(lambda: ops.conv3x3_wgrad_wino(x, dz, cout, dz_idx=idx))
    print("%s wgrad wino %6.1f us (%5.1f TF)" % (name, t_ww, gf / t_ww * 1e3))
    print("%s %3dx%-3d %3d->%-3d  fwd direct %6.1f us (%5.1f TF)  wino %6.1f us (%5.1f TF) | dgrad direct %6.1f (%5.1f)  wino %6.1f (%5.1f) | wgrad %6.1f (%5.1f)" % (
        name, hw, hw, cin, cout, t_fd, gf / t_fd * 1e3, t_fw, gf / t_fw * 1e3, t_dd, gf / t_dd * 1e3, t_dw, gf / t_dw * 1e3, t_wg, gf / t_wg * 1e3), flush=True)
