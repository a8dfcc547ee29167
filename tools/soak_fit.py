#!/usr/bin/env python3
"""Soak of the batch pipeline behind `GaitSetModel.fit` (ugaitnet_amd/keras_compat.py _BatchPipeline): 3 epochs x 40 steps of 8-clip C3
batches from a generator with random host delays, (a) plain loop, (b) one background fetcher + staging + late loss read, (c) four
ordered fetchers with a short queue -- History and every parameter must come out bit-identical.  python tools/soak_fit.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from tests.synth import make_batch
from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet3Mods, optimizers, sign_max
class Gen:
    def __init__(self, b, nb, jitter):
        self.jitter = jitter
        self.batches = []
        for i in range(nb):
            xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), b, 25, 150, ids=b // 2, seed=500 + i)
            self.batches.append(([a for pair in zip(xs, uses) for a in pair], [labels.reshape(-1, 1).astype(np.float32), onehot]))
        self.rng = np.random.default_rng(0)
    def __len__(self): return len(self.batches)
    def __getitem__(self, i):
        X, Y = self.batches[i]
        if self.jitter: time.sleep(float(self.rng.uniform(0, 0.02)))
        return [a.copy() for a in X], Y
    def on_epoch_end(self): pass
shapes = [(25, 60, 60, 2), (25, 60, 60, 1), (25, 60, 60, 1)]
res = {}
for name, kw, jit in (("plain", dict(workers=0, pipeline=False), False), ("piped", dict(), True), ("pool", dict(workers=4, max_queue_size=3), True)):
    model = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], ndense_units=0, optimizer=optimizers.Adam(lr=1e-4),
                                           margin=0.2, nclasses=150, loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=1)
    t0 = time.time()
    h = model.fit(Gen(8, 10, jit), epochs=3, steps_per_epoch=40, verbose=0, **kw)
    torch.cuda.synchronize()
    res[name] = (h.history, {n: model.core.store.get(n).copy() for n in model.core.store.names})
    print(name, "%.1f s" % (time.time() - t0), h.history["loss"], flush=True)
for name in ("piped", "pool"):
    assert res[name][0] == res["plain"][0], name
    for n, w in res["plain"][1].items():
        assert np.array_equal(w, res[name][1][n]), (name, n)
print("120 steps: History and every parameter bit-identical in the three modes")
