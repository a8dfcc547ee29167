"""The encoder of nets/mj_uwyhNets_ba.py:419-484 with its 3x3 layers on the f16 matrix pipe: kernel sequencing for
`GaitCore(conv_precision="h2")`.

Between the 5x5 first layer and HPP every activation and gradient is an H2 tensor (ugaitnet_amd/csrc/mm_common.h: two f16
halves per element + a block exponent per tensor; the bytes of the fp32 tensor it replaces).  The arithmetic is the fp32
path's, step for step (engine.forward_merged / backward_merged): what changes is the format between the kernels and the
kernels that read it.  One launch per layer for all modalities (frame-level layer + set-level twin as up to six jobs).
"""
from __future__ import annotations

import os

import torch

from . import h2, ops

F32 = torch.float32
U8 = torch.uint8
I16 = torch.int16
I32 = torch.int32
NBINS, FEAT, HIDDEN = 62, 128, 256
# Settings.fuse_w5 (UGN_FUSE_W5=1): the a2 data gradient multiplies its registers with the input patch in place and dL/da1 is never written
# (dgrad32_w5_kernel).  Correct (tests/test_engine_gpu.py) but not the default: the fused launch takes 704 us against 469 + 278 for
# the two it replaces, and those 278 ran on the second stream -- the step got 0.12 ms LONGER (the patch gather and the on-the-fly
# split of both operands cost more vector instructions than the stores they replace).
# Settings.set_routed (UGN_SET_ROUTED=0): the set-pooling gradients find the maximum frames by reading the frames again (the round-3
# kernels; same results).  Both switches are read from the owning core's settings (enc.cfg, ugaitnet_amd/config.py).
# 3x3 layers: name, cin, cout, spatial size, pooled
LAYERS3 = (("a2", 32, 32, 64, True), ("b1", 32, 64, 32, False), ("b2", 64, 64, 32, True), ("a3", 32, 64, 32, False),
           ("a4", 64, 64, 32, True), ("b3", 64, 128, 16, False), ("b4", 128, 128, 16, False), ("a5", 64, 128, 16, False),
           ("a6", 128, 128, 16, False))


class H2State:
    """What one modality branch holds for the H2 path: packed filter halves (+ their exponents / L1 bounds) and the cache of
    H2 activation / gradient buffers, whose metas live in the model's MetaPool."""

    def __init__(self, enc, pool):
        self.enc, self.pool = enc, pool
        dev = enc.store.device
        self.wmeta = torch.zeros((len(LAYERS3), 2, 4), dtype=I32, device=dev)     # [layer][direction] ugn_wmeta: NOT reset per step
        self.pk = {}
        for k, (name, cin, cout, _, _) in enumerate(LAYERS3):
            self.pk[name] = tuple((torch.empty((18 * cin * cout,), dtype=I16, device=dev), self.wmeta[k, d]) for d in (0, 1))
        self.bufs = {}

    def pack_jobs(self):
        jobs = [(self.enc.W(name), self.pk[name][d][0], self.pk[name][d][1], d) for name, *_ in LAYERS3 for d in (0, 1)]
        if self.enc.cfg.fuse_w5:     # the fused kernel reads the a2 data-gradient filter in the 32-column block layout: a copy of its own
            if "a2w5" not in self.pk:
                dev = self.enc.store.device
                self.pk["a2w5"] = (torch.empty((18 * 32 * 32,), dtype=I16, device=dev), torch.zeros(4, dtype=I32, device=dev))
            jobs.append((self.enc.W("a2"), self.pk["a2w5"][0], self.pk["a2w5"][1], 3))
        return jobs

    def wf(self, name):
        return self.pk[name][0]

    def wd(self, name):
        return self.pk[name][1]

    def t(self, key, shape):
        """H2 buffer `key` of logical NHWC `shape`; its meta is the pool record '<prefix><key>'."""
        buf = self.bufs.get(key)
        if buf is None or buf.shape != tuple(shape):
            buf = self.bufs[key] = h2.H2Tensor.empty(tuple(shape), self.enc.store.device, meta=self.pool.slot(self.enc.prefix + key))
        return buf

    def alias(self, key, other):
        """An H2Tensor over `other`'s data with the meta record '<prefix><key>' (in-place results get their own exponent)."""
        return h2.H2Tensor(other.data, self.pool.slot(self.enc.prefix + key))

    def f32(self, key, shape, dtype=F32):
        buf = self.bufs.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = self.bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=self.enc.store.device)
        return buf

    def slot(self, key):
        return self.pool.slot(self.enc.prefix + key)


def forward_h2(encs, xs):
    """[x_m [B_m,L,60,60,C_m]] -> [[62,B_m,256]] for the branches `encs` (each with .h2 = H2State) in lockstep."""
    S = [e.h2 for e in encs]
    geo = []
    for e, x in zip(encs, xs):
        b, l = x.shape[0], x.shape[1]
        e.shape = (b, l)
        geo.append((b, l, b * l))
    k = len(encs)
    R = range(k)
    bs, l0 = [g[0] for g in geo], geo[0][1]
    assert all(g[1] == l0 for g in geo), "the modalities of a batch share the set length"
    # block 1: max|x| of every modality (the first layer's exponent), the 5x5 layer per modality, a2 for all of them
    xfs = [x.reshape(g[2], 60, 60, e.cin) for e, x, g in zip(encs, xs, geo)]
    for s, xf in zip(S, xfs):
        s.x = xf
    h2.absmax_multi(xfs, [s.slot("x") for s in S])
    a1s = [h2.conv5x5_in_fwd_h2(xfs[i], S[i].slot("x"), encs[i].W("a1"), S[i].t("a1", (geo[i][2], 64, 64, 32)),
                                sign=S[i].f32("a1s", (geo[i][2], 64, 64), I32)) for i in R]
    hook = getattr(encs[0], "before_conv3", None)     # (the packed 3x3 filters: repacked beside the 5x5 layer, engine.apply_gradients)
    if hook is not None:
        hook()
    p2s = [S[i].t("p2", (geo[i][2], 32, 32, 32)) for i in R]
    i2s = [S[i].f32("i2", (geo[i][2], 32, 32, 32), U8) for i in R]
    h2.conv3x3_fwd_mm_multi(a1s, [s.wf("a2")[0] for s in S], [s.wf("a2")[1] for s in S], 32, True, p2s, i2s)
    m1s = [S[i].t("m1", (geo[i][0], 32, 32, 32)) for i in R]
    # routing words of the three set poolings (which frames hold the maximum / are positive): their gradients read these 8 bytes
    # per set element instead of the l frames again
    routes = lambda key, hw, c: [S[i].f32(key, (geo[i][0], hw, hw, 2, c), I32) for i in R] if encs[0].cfg.set_routed and l0 <= 32 else None
    h2.setmax_fwd_h2_multi(p2s, bs, l0, ms=m1s, routes=routes("r1", 32, 32))

    def pair_layer(na, nb, xa, xb, cout, hw, pool, ka, kb, ia=None, ib=None):
        """frame-level layer `na` on xa and its set-level twin `nb` on xb for every modality: jobs = [frame..., set...]"""
        ho = hw // 2 if pool else hw
        outs = [S[i].t(ka, (geo[i][2], ho, ho, cout)) for i in R] + [S[i].t(kb, (geo[i][0], ho, ho, cout)) for i in R]
        idxs = None
        if pool:
            idxs = [S[i].f32(ia, (geo[i][2], ho, ho, cout), U8) for i in R] + [S[i].f32(ib, (geo[i][0], ho, ho, cout), U8) for i in R]
        h2.conv3x3_fwd_mm_multi(list(xa) + list(xb), [s.wf(na)[0] for s in S] + [s.wf(nb)[0] for s in S],
                                [s.wf(na)[1] for s in S] + [s.wf(nb)[1] for s in S], cout, pool, outs, idxs)
        return outs[:k], outs[k:]

    a3s, b1s = pair_layer("a3", "b1", p2s, m1s, 64, 32, False, "a3", "b1")
    p4s, q2s = pair_layer("a4", "b2", a3s, b1s, 64, 32, True, "p4", "q2", "i4", "j2")
    s2s = [S[i].t("s2", (geo[i][0], 16, 16, 64)) for i in R]
    h2.setmax_fwd_h2_multi(p4s, bs, l0, addends=q2s, sums=s2s, routes=routes("r2", 16, 64))
    a5s, b3s = pair_layer("a5", "b3", p4s, s2s, 128, 16, False, "a5", "b3")
    a6s, b4s = pair_layer("a6", "b4", a5s, b3s, 128, 16, False, "a6", "b4")
    # the last set pooling leaves the H2 part of the path: HPP and the per-bin FC stay fp32
    m3s = [S[i].f32("m3", (geo[i][0], 16, 16, 128)) for i in R]
    s3s = [S[i].f32("s3", (geo[i][0], 16, 16, 128)) for i in R]
    h2.setmax_fwd_h2_f32_multi(a6s, bs, l0, m3s, b4s, s3s, routes=routes("r3", 16, 128))
    feats = ops.hpp_fwd_multi(m3s, s3s, [S[i].f32("feat", (NBINS, geo[i][0], FEAT)) for i in R])
    outs = ops.binfc_fwd_multi(feats, [e.W("fc") for e in encs], [S[i].f32("out", (NBINS, geo[i][0], HIDDEN)) for i in R])
    for e, o in zip(encs, outs):
        e.act = {"out": o}      # (what callers of the fp32 path read from an encoder)
    return outs


def backward_h2(encs, douts, side):
    """Backward of forward_h2: parameter gradients into the store's grad views.  `side(device)` is the context manager that
    runs the enclosed launches on the weight-gradient stream (engine._side)."""
    S = [e.h2 for e in encs]
    dev = encs[0].store.device
    geo = [(e.shape[0], e.shape[1], e.shape[0] * e.shape[1]) for e in encs]
    k = len(encs)
    R = range(k)
    bs, l0 = [g[0] for g in geo], geo[0][1]
    T = lambda key: [s.bufs[key] for s in S]
    RT = lambda key: T(key) if encs[0].cfg.set_routed and l0 <= 32 else None       # routing words of a set pooling (forward_h2)
    fc_args = (T("feat"), [e.W("fc") for e in encs], douts, [e.G("fc") for e in encs],
                                    [S[i].f32("dfeat", (NBINS, geo[i][0], FEAT)) for i in R])
    ops.binfc_bwd_multi(*fc_args, parts=2)            # dfeat: the rest of the backward pass waits for it
    with side(encs[0].store.device):
        ops.binfc_bwd_multi(*fc_args, parts=1)        # the FC weight gradients: on the second stream
    dfeats = fc_args[4]
    dm3s = [S[i].f32("dm3", (geo[i][0], 16, 16, 128)) for i in R]
    dzb4f = [S[i].f32("dzb4f", (geo[i][0], 16, 16, 128)) for i in R]
    h2.hpp_bwd_b4h2_multi(T("m3"), T("s3"), T("b4"), dfeats, dm3s, dzb4f)
    # the two gradients enter the H2 part of the path: their maxima (the exponents), then dzb4 as H2; dm3 stays fp32 -- the
    # set-max gradient reads it directly
    h2.absmax_multi(dm3s + dzb4f, [s.slot("dm3") for s in S] + [s.slot("dzb4f") for s in S])
    dzb4 = h2.encode_multi(dzb4f, [s.slot("dzb4f") for s in S], [S[i].t("dzb4", (geo[i][0], 16, 16, 128)) for i in R])
    dz6 = h2.setmax_bwd_h2_multi(T("a6"), dm3s, [s.slot("dm3") for s in S], bs, l0, True,
                                 [S[i].t("dz6", (geo[i][2], 16, 16, 128)) for i in R], dm_is_f32=True, routes=RT("r3"))

    def wgrad(na, nb, xa, xb, dza, dzb, cout, ia=None, ib=None):
        with side(dev):
            h2.conv3x3_wgrad_mm_multi(list(xa) + list(xb), list(dza) + list(dzb), cout, [e.G(na) for e in encs] + [e.G(nb) for e in encs],
                                      dz_idxs=None if ia is None else list(ia) + list(ib))

    def dgrad(na, nb, dza, dzb, hw, cin, cout, ka, kb, ia=None, ib=None, acta=None, actb=None):
        outa = [S[i].t(ka, (geo[i][2], hw, hw, cin)) for i in R]
        outb = [S[i].t(kb, (geo[i][0], hw, hw, cin)) for i in R]
        h2.conv3x3_dgrad_mm_multi(list(dza) + list(dzb), [s.wd(na)[0] for s in S] + [s.wd(nb)[0] for s in S],
                                  [s.wd(na)[1] for s in S] + [s.wd(nb)[1] for s in S], hw, cin, cout, outa + outb,
                                  dz_idxs=None if ia is None else list(ia) + list(ib),
                                  acts=None if acta is None else list(acta) + list(actb))
        return outa, outb

    # block 3 of the frame stack (a5, a6) with block 2 of the global branch (b3, b4)
    wgrad("a6", "b4", T("a5"), T("b3"), dz6, dzb4, 128)
    dz5, dzb3 = dgrad("a6", "b4", dz6, dzb4, 16, 128, 128, "dz5", "dzb3", acta=T("a5"), actb=T("b3"))
    wgrad("a5", "b3", T("p4"), T("s2"), dz5, dzb3, 128)
    # plain epilogue for both data gradients of the pair: the set-max backward adds the frame-level extras (+ set-max gradient of
    # p4, * LeakyReLU'(p4)), a small elementwise kernel the set-level one (* LeakyReLU'(q2))
    raw4, ds2 = dgrad("a5", "b3", dz5, dzb3, 16, 64, 128, "g4", "ds2")
    dq2 = h2.lrelu_bwd_h2_multi(ds2, T("q2"), [S[i].t("dq2", (geo[i][0], 16, 16, 64)) for i in R])
    dp4 = h2.setmax_bwd_h2_multi(T("p4"), ds2, [d.meta for d in ds2], bs, l0, True, [S[i].alias("dp4", raw4[i]) for i in R], addends=raw4,
                                 routes=RT("r2"))
    # block 2 (a3, a4) with block 1 of the global branch (b1, b2); a4 / b2 are pooled
    wgrad("a4", "b2", T("a3"), T("b1"), dp4, dq2, 64, T("i4"), T("j2"))
    dz3, dzb1 = dgrad("a4", "b2", dp4, dq2, 32, 64, 64, "dz3", "dzb1", T("i4"), T("j2"), T("a3"), T("b1"))
    wgrad("a3", "b1", T("p2"), T("m1"), dz3, dzb1, 64)
    raw2, dm1 = dgrad("a3", "b1", dz3, dzb1, 32, 32, 64, "g2", "dm1")
    dp2 = h2.setmax_bwd_h2_multi(T("p2"), dm1, [d.meta for d in dm1], bs, l0, True, [S[i].alias("dp2", raw2[i]) for i in R], addends=raw2,
                                 routes=RT("r1"))
    # block 1 (a1, a2): dz1 = dL/da1; the first layer's LeakyReLU' comes from its sign bits inside the 5x5 weight gradient
    i2 = T("i2")
    with side(dev):
        h2.conv3x3_wgrad_mm_multi(T("a1"), dp2, 32, [e.G("a2") for e in encs], dz_idxs=i2)
    if encs[0].cfg.fuse_w5:
        # dL/da1 has one consumer, the 5x5 layer's weight gradient: the a2 data gradient multiplies its registers with the input
        # patch in place (dgrad32_w5_kernel) -- 0.94 GB per step neither written nor read back, three launches fewer
        h2.dgrad32_wgrad5_multi(dp2, i2, [s.pk["a2w5"][0] for s in S], [s.pk["a2w5"][1] for s in S], [s.x for s in S],
                                [s.slot("x") for s in S], [s.bufs["a1s"] for s in S], [e.G("a1") for e in encs],
                                [s.slot("w5scale") for s in S])
        return
    dz1 = [S[i].t("dz1", (geo[i][2], 64, 64, 32)) for i in R]
    h2.conv3x3_dgrad_mm_multi(dp2, [s.wd("a2")[0] for s in S], [s.wd("a2")[1] for s in S], 64, 32, 32, dz1, dz_idxs=i2)
    with side(dev):
        for i, e in enumerate(encs):
            h2.conv5x5_in_wgrad_h2(S[i].x, dz1[i], e.G("a1"), sign=S[i].bufs["a1s"], x_meta=S[i].slot("x"))
