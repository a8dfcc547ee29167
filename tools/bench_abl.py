import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops
dev = torch.device("cuda")
N, hw, cin, cout = 600, 16, 128, 128
x = torch.randn(N, hw, hw, cin, device=dev); w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
uf = ops.wino_pack(w, False)
for _ in range(3): ops.conv3x3_fwd_wino(x, uf, cout, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv3x3_fwd_wino(x, uf, cout, False)
e1.record(); torch.cuda.synchronize()
print("UGN_ABL=%s  %.1f us" % (os.environ.get("UGN_ABL", "0"), e0.elapsed_time(e1) * 100))
