"""Forward-only (evaluation) throughput: signatures of 64-clip batches under all-modalities input, as the reference's test main
encodes its gallery / probe sets (mains/mj_testUWYHGaitNet_open_tum.py:139-224, batch 64).  usage: bench_forward.py [f32|bf16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.synth import make_batch
from ugaitnet_amd.engine import GaitCore

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
b = 64
xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), b, 25, 150, seed=232323, masks=False)
dxs = [torch.from_numpy(x).cuda() for x in xs]
dus = [torch.from_numpy(u).cuda() for u in uses]
core = GaitCore([2, 1, 1], nclasses=150, fuse_mode="sign_max", seed=1, conv_precision=prec)
for _ in range(3):
    core.predict(dxs, dus)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    core.predict(dxs, dus)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(metric="clips/sec (3-mod, L=25, 60x60) forward only (signature + classprob)", dtype=prec, clips_per_batch=b,
                      ms_per_batch=round(dt * 1e3, 3), value=round(b / dt, 1), unit="clips/s")))
