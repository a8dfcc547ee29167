import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.synth import make_batch
from ugaitnet_amd import engine
from ugaitnet_amd.engine import GaitCore
xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), 8, 6, 20, ids=4, seed=232323)
dxs = [torch.from_numpy(x).cuda() for x in xs]; dus = [torch.from_numpy(u).cuda() for u in uses]; doh = torch.from_numpy(onehot).cuda()
def run(prec, **cfg):
    core = GaitCore([2, 1, 1], nclasses=20, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=232323, lr=1e-3,
                    conv_precision=prec, config=engine.DEFAULTS.replace(**cfg))
    out = []
    for s in range(12):
        core.train_step(dxs, dus, labels, doh)
        out.append(core.losses()["loss"])
    torch.cuda.synchronize()
    return out, core.store.flat.cpu().numpy().copy()
a, pa = run("f32x3")
b, pb = run("f32x3", pack_on_side_stream=False)
c, pc = run("f32")
d, pd = run("f32x3", wgrad_stream=False)
print("x3 side-pack == main-pack bitwise:", np.array_equal(pa, pb), " == no side streams:", np.array_equal(pa, pd))
for i in range(12): print(i, "x3 %.6f  x3(main pack) %.6f  f32 %.6f  x3-f32 %+.2e" % (a[i], b[i], c[i], a[i] - c[i]))
