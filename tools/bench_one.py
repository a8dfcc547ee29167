"""Launch each Winograd kernel of ONE layer shape a few times (PMC / trace target). usage: bench_one.py a6 [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops

CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}
hw, cin, cout, pool = CFGS[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 600
dev = torch.device("cuda")
x = torch.randn(N, hw, hw, cin, device=dev)
w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
uf, ud = ops.wino_pack(w, False), ops.wino_pack(w, True, pooled_dz=pool)
ho = hw // 2 if pool else hw
dz = torch.randn(N, ho, ho, cout, device=dev)
idx = torch.randint(0, 4, (N, ho, ho, cout), device=dev, dtype=torch.uint8) if pool else None
act = torch.randn(N, hw, hw, cin, device=dev)
for _ in range(reps):
    ops.conv3x3_fwd_wino(x, uf, cout, pool)
    ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act)
    ops.conv3x3_wgrad_wino(x, dz, cout, dz_idx=idx)
torch.cuda.synchronize()
