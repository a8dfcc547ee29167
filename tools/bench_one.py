"""Run one Winograd forward shape repeatedly (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops
hw, cin, cout, pool = (int(a) for a in sys.argv[1:5])
N = 600
dev = torch.device("cuda")
x = torch.randn(N, hw, hw, cin, device=dev)
w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
uf = ops.wino_pack(w, False)
wp = ops.pack3x3(w)
for _ in range(3):
    ops.conv3x3_fwd_wino(x, uf, cout, bool(pool))
    ops.conv3x3_fwd(x, wp, bool(pool))
torch.cuda.synchronize()
