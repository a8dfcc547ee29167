"""Cost of the data-gradient epilogue operands: dgrad with / without act (LeakyReLU' mask) and addend, per layer shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops

CFGS = [(64, 32, 32, True, "a2"), (32, 32, 64, False, "a3"), (32, 64, 64, True, "a4"), (16, 64, 128, False, "a5"), (16, 128, 128, False, "a6")]
N = 600
dev = torch.device("cuda")


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


for hw, cin, cout, pool, name in CFGS:
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    ud = ops.wino_pack(w, True, pooled_dz=pool)
    ho = hw // 2 if pool else hw
    dz = torch.randn(N, ho, ho, cout, device=dev)
    idx = torch.randint(0, 4, (N, ho, ho, cout), device=dev, dtype=torch.uint8) if pool else None
    act = torch.randn(N, hw, hw, cin, device=dev)
    add = torch.randn(N, hw, hw, cin, device=dev)
    out = torch.empty(N, hw, hw, cin, device=dev)
    t0 = timeit(lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, out=out))
    t1 = timeit(lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act, out=out))
    t3 = timeit(lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act, addend=add, out=out))
    print("%s dgrad plain %6.1f us | +act %6.1f | +act+addend %6.1f   (out %.0f MB)" % (name, t0, t1, t3, out.numel() * 4 / 1e6), flush=True)
