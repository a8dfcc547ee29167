"""Time of the per-step filter repack of one branch (ugn_wino_pack_multi, 18 jobs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ugaitnet_amd import ops

dev = torch.device("cuda")
shapes = [(32, 32, True), (32, 64, False), (64, 64, True), (64, 128, False), (128, 128, False)] * 2
jobs = []
for cin, cout, pool in shapes[:9]:
    w = torch.randn(3, 3, cin, cout, device=dev)
    for dg in (False, True):
        jobs.append((w, torch.empty(16 * cin * cout, device=dev), dg, pool))
for _ in range(3):
    ops.wino_pack_multi(jobs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.wino_pack_multi(jobs)
e1.record()
torch.cuda.synchronize()
print("wino_pack_multi, %d jobs: %.1f us" % (len(jobs), e0.elapsed_time(e1) / 20 * 1e3))
