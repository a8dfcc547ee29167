"""Race check for the overlapped backward: N training steps with and without the side stream must give bit-identical
parameters (every kernel is deterministic, so any difference would be an ordering hazard)."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import torch, numpy as np
from tests.synth import make_batch
from ugaitnet_amd.engine import GaitCore
steps = int(sys.argv[1])
core = GaitCore([2, 1, 1], nclasses=150, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=3, lr=1e-3)
h = hashlib.sha256()
for s in range(steps):
    xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), 24, 25, 150, seed=100 + s %% 4)
    core.train_step([torch.from_numpy(x).cuda() for x in xs], [torch.from_numpy(u).cuda() for u in uses], labels,
                    torch.from_numpy(onehot).cuda())
    if s %% 5 == 4:
        h.update(core.store.flat.cpu().numpy().tobytes())
print(h.hexdigest(), core.losses()["loss"])
''' % ROOT

if __name__ == "__main__":
    steps = sys.argv[1] if len(sys.argv) > 1 else "20"
    # Two launch shapes -- one launch per layer for all modalities (default) and one per layer and modality (UGN_MERGE=0) -- sum
    # the weight gradients in different orders, so each is compared with itself under its stream modes.
    groups = (({"UGN_WSTREAM": "0"}, {"UGN_WSTREAM": "1"}),
              # (one launch per layer and modality exists on the Winograd fp32-MFMA set only: the default x3 set runs merged launches)
              ({"UGN_CONV_PRECISION": "f32", "UGN_MERGE": "0", "UGN_WSTREAM": "0", "UGN_FSTREAMS": "0"},
               {"UGN_CONV_PRECISION": "f32", "UGN_MERGE": "0", "UGN_WSTREAM": "1", "UGN_FSTREAMS": "2"},
               {"UGN_CONV_PRECISION": "f32", "UGN_MERGE": "0", "UGN_WSTREAM": "1", "UGN_FSTREAMS": "2", "UGN_BSTREAMS": "1"}))
    for envs in groups:
        outs = []
        for env in envs:
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, "-c", CHILD, steps], env=e, capture_output=True, text=True)
            print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
            outs.append(r.stdout.strip().splitlines()[-1].split()[0] if r.stdout.strip() else None)
        assert outs[0] is not None and all(o == outs[0] for o in outs), "parameter trajectories differ between stream modes"
    print("identical trajectories over %s steps" % steps)
