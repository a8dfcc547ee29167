#!/usr/bin/env python3
"""End-to-end rate of the Keras-surface training loop (`GaitSetModel.fit`, the loop nets/mj_uwyhNets_ba.py:963 drives) on C3 batches
that arrive as HOST numpy arrays from a keras.utils.Sequence-shaped generator -- generator call, PCIe transfer, step, loss readback --
against bench.py's device-resident step.  Prints one line per mode:
    plain   workers=0, pipeline=False: fetch, copy, step, read the losses, one after the other
    piped   the defaults (workers=1, max_queue_size=10, losses one step late): what tf.keras's fit does for a Sequence
usage: python tools/bench_fit.py [--clips 24] [--steps 30] [--host-ms 0] [--workers 1]   (--host-ms: extra host work per generator call)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=24)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--host-ms", type=float, default=0.0)
    ap.add_argument("--workers", type=int, default=1)
    args = ap.parse_args()
    import torch
    from tests.synth import make_batch
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet3Mods, optimizers, sign_max

    class Gen:
        def __init__(self):
            self.batches = []
            self.scratch = np.zeros((2, 1 << 20), np.float32)
            for i in range(4):
                xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), args.clips, 25, 150, ids=args.clips // 2, seed=232323 + i)
                self.batches.append(([a for pair in zip(xs, uses) for a in pair], [labels.reshape(-1, 1).astype(np.float32), onehot]))

        def __len__(self):
            return len(self.batches)

        def __getitem__(self, i):
            X, Y = self.batches[i]
            t0 = time.perf_counter()
            X = [a.copy() for a in X]          # a generator hands out fresh arrays (the reference's assembles them from HDF5 samples)
            while (time.perf_counter() - t0) * 1e3 < args.host_ms:      # stand-in for the reference generator's numpy / HDF5 work:
                np.copyto(self.scratch[1], self.scratch[0])               # array calls that release the GIL, not a Python spin
            return X, Y

        def on_epoch_end(self):
            pass

    shapes = [(25, 60, 60, 2), (25, 60, 60, 1), (25, 60, 60, 1)]
    for name, kw in (("plain", dict(workers=0, pipeline=False)), ("piped", dict(workers=args.workers)), ("plain", dict(workers=0, pipeline=False)),
                     ("piped", dict(workers=args.workers))):
        model = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], ndense_units=0, optimizer=optimizers.Adam(lr=1e-4),
                                               margin=0.2, nclasses=150, loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=1)
        gen = Gen()
        model.fit(gen, epochs=1, steps_per_epoch=5, verbose=0, **kw)       # warm-up: buffers, pinned pool, filter pack
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = model.fit(gen, epochs=1, steps_per_epoch=args.steps, verbose=0, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s: %.3f ms per step, %.0f clips/s  (loss %.5f; %d clips, host work %.1f ms per batch)" %
              (name, dt / args.steps * 1e3, args.clips * args.steps / dt, hist.history["loss"][0], args.clips, args.host_ms), flush=True)
        del model


if __name__ == "__main__":
    main()
