#!/usr/bin/env python3
"""Re-run single launches of the H2 training step on the tensors the step itself produced (real activations, real exponents),
and on random data of the same shapes -- to separate what the DATA costs from what the schedule around a launch costs.

    python tools/bench_step_kernels.py [--reps 10]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tests.synth import make_batch
from ugaitnet_amd import h2
from ugaitnet_amd.engine import GaitCore


def main():
    argv = sys.argv[1:]
    reps = int(argv[argv.index("--reps") + 1]) if "--reps" in argv else 10
    dev = torch.device("cuda")
    kinds = ("of", "gray", "depth")
    xs, uses, labels, onehot = make_batch(kinds, 24, 25, 150, ids=12, seed=232323)
    core = GaitCore([2, 1, 1], nclasses=150, loss_weights=(1.0, 0.1), seed=1, conv_precision="h2")
    dxs = [torch.from_numpy(x).to(dev) for x in xs]
    dus = [torch.from_numpy(u).to(dev) for u in uses]
    doh = torch.from_numpy(onehot).to(dev)
    for _ in range(2):
        core.forward_backward(dxs, dus, labels, doh)
    torch.cuda.synchronize()
    S = [e.h2 for e in core.encoders]
    T = lambda key: [s.bufs[key] for s in S]
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)

    def timeit(fn, cold):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            if cold:
                junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ts]) * 1e3)

    def stats(ts, name):
        for t in ts:
            v = t.numpy()
            e, bits = t.meta.cpu().numpy().tolist()
            amax = float(np.array([bits], np.uint32).view(np.float32)[0])
            st = np.abs(v) * 2.0 ** e
            nz = st[st > 0]
            print("    %-6s n=%4d e=%4d stored amax %.3g  median |stored| %.3g  zeros %.1f %%  |stored| < 2^-14: %.1f %%  < 2^-3: %.1f %%"
                  % (name, v.shape[0], e, amax, float(np.median(nz)) if nz.size else 0.0, 100.0 * (st == 0).mean(),
                     100.0 * ((st > 0) & (st < 2.0 ** -14)).mean(), 100.0 * ((st > 0) & (st < 2.0 ** -3)).mean()))

    def fwd(na, nb, xa, xb, cout, pool, label):
        ins = list(xa) + list(xb)
        wf = [s.wf(na) for s in S] + [s.wf(nb) for s in S]
        n_, hw, _, cin = ins[0].shape
        ho = hw // 2 if pool else hw
        outs = [h2.H2Tensor.empty((x.shape[0], ho, ho, cout), dev) for x in ins]
        idxs = [torch.empty((x.shape[0], ho, ho, cout), dtype=torch.uint8, device=dev) for x in ins] if pool else None
        run = lambda xs_: h2.conv3x3_fwd_mm_multi(xs_, [w[0] for w in wf], [w[1] for w in wf], cout, pool, outs, idxs)
        rnd = [h2.encode(torch.randn(x.shape, device=dev)) for x in ins]
        print("%-10s real data: %7.1f us hot %7.1f us cold | random data: %7.1f us hot %7.1f us cold" % (
            label, timeit(lambda: run(ins), False), timeit(lambda: run(ins), True), timeit(lambda: run(rnd), False), timeit(lambda: run(rnd), True)))
        stats(ins[:1] + ins[3:4], "in")

    fwd("a3", "b1", T("p2"), T("m1"), 64, False, "a3|b1 fwd")
    fwd("a4", "b2", T("a3"), T("b1"), 64, True, "a4|b2 fwd")
    fwd("a5", "b3", T("p4"), T("s2"), 128, False, "a5|b3 fwd")
    fwd("a6", "b4", T("a5"), T("b3"), 128, False, "a6|b4 fwd")


if __name__ == "__main__" and "--in-step" not in sys.argv:
    main()


def in_step():
    """Every forward 3x3 launch of a real step timed in place, then issued a SECOND time right behind itself (same tensors):
    if the repeat is much faster, the first pays for the state the previous kernels left (caches, TLB, dirty lines)."""
    dev = torch.device("cuda")
    kinds = ("of", "gray", "depth")
    xs, uses, labels, onehot = make_batch(kinds, 24, 25, 150, ids=12, seed=232323)
    core = GaitCore([2, 1, 1], nclasses=150, loss_weights=(1.0, 0.1), seed=1, conv_precision="h2")
    dxs = [torch.from_numpy(x).to(dev) for x in xs]
    dus = [torch.from_numpy(u).to(dev) for u in uses]
    doh = torch.from_numpy(onehot).to(dev)
    from ugaitnet_amd import engine
    orig = h2.conv3x3_fwd_mm_multi
    rec, rec2 = [], []

    fresh = {}

    def timed(xs_, wpks, wmetas, cout, pool, outs, idxs=None):
        key = (xs_[0].shape, cout)
        if key not in fresh:      # output tensors / input copies of this launch that the engine does not own
            fresh[key] = ([h2.H2Tensor.empty(o.shape, dev) for o in outs],
                          [torch.empty_like(i) for i in idxs] if pool else None,
                          [h2.H2Tensor(x.data.clone(), x.meta.clone()) for x in xs_])
        fo, fi, fx = fresh[key]
        for a, b in zip(fx, xs_):
            a.data.copy_(b.data)
            a.meta.copy_(b.meta)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        orig(xs_, wpks, wmetas, cout, pool, outs, idxs)
        ev[1].record()
        orig(xs_, wpks, wmetas, cout, pool, outs, idxs)
        ev[2].record()
        orig(xs_, wpks, wmetas, cout, pool, fo, fi)
        ev[3].record()
        orig(fx, wpks, wmetas, cout, pool, outs, idxs)
        ev[4].record()
        # the engine's DATA on both sides, but metas outside the model's MetaPool (one 8-byte record per tensor, side by side)
        xm = [h2.H2Tensor(x.data, f.meta) for x, f in zip(xs_, fx)]
        om = [h2.H2Tensor(o.data, f.meta) for o, f in zip(outs, fo)]
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev2[0].record()
        orig(xm, wpks, wmetas, cout, pool, outs, idxs)
        ev2[1].record()
        orig(xs_, wpks, wmetas, cout, pool, om, idxs)
        ev2[2].record()
        orig(xm, wpks, wmetas, cout, pool, om, idxs)
        ev2[3].record()
        rec2.append(ev2)
        rec.append(("%d->%d @%d" % (xs_[0].shape[3], cout, xs_[0].shape[1]), ev))
        return (outs, idxs) if pool else outs
    h2.conv3x3_fwd_mm_multi = timed
    with core.serial_launches():
        for _ in range(4):
            rec.clear()
            rec2.clear()
            core.forward_backward(dxs, dus, labels, doh)
        torch.cuda.synchronize()
    for name, ev in rec:
        print("in step  %-14s first %7.1f us   repeated at once %7.1f us   other OUTPUT tensors %7.1f us   copies of the INPUT tensors %7.1f us"
              % (name, ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3, ev[3].elapsed_time(ev[4]) * 1e3))
    for (name, _), ev2 in zip(rec, rec2):
        print("in step  %-14s engine data, input metas outside the pool %7.1f us   output metas outside %7.1f us   both outside %7.1f us"
              % (name, ev2[0].elapsed_time(ev2[1]) * 1e3, ev2[1].elapsed_time(ev2[2]) * 1e3, ev2[2].elapsed_time(ev2[3]) * 1e3))


if "--in-step" in sys.argv:
    in_step()
