"""Loss trajectory of the same job (C3 batch, fixed) in the default fp32-tensor arithmetic (x3), on the Winograd fp32-MFMA kernels, in f16x2 and in bf16: do they train alike?
usage: train_curve.py <steps> <out.json> [f32x3,f32,h2,bf16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.synth import make_batch
from ugaitnet_amd.engine import GaitCore

steps, out = int(sys.argv[1]), sys.argv[2]
xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), 24, 25, 150, seed=232323)
dxs = [torch.from_numpy(x).cuda() for x in xs]
dus = [torch.from_numpy(u).cuda() for u in uses]
doh = torch.from_numpy(onehot).cuda()
res = {}
for prec in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("f32x3", "f32", "h2", "bf16")):
    core = GaitCore([2, 1, 1], nclasses=150, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=232323, lr=1e-4,
                    conv_precision=prec)
    curve = []
    for s in range(steps):
        core.train_step(dxs, dus, labels, doh)
        if s % 10 == 0 or s == steps - 1:
            l = core.losses()
            curve.append(dict(step=s, loss=round(l["loss"], 5), triplet=round(l["triplet"], 5), xent=round(l["xent"], 5), acc=l["acc"]))
    res[prec] = curve
    print(prec, curve[0]["loss"], "->", curve[-1]["loss"], "acc", curve[-1]["acc"], flush=True)
json.dump(dict(note="same fixed C3 batch (24 clips, 3 modalities, masks), same initial weights, Adam 1e-4; loss = triplet + 0.1 xent", curves=res),
          open(out, "w"), indent=1)
