"""The encoder of nets/mj_uwyhNets_ba.py:419-484 with bf16 tensors in HBM: kernel sequencing for `GaitCore(conv_precision="bf16")`
-- BASELINE.json configs[4] ("3-modality bf16 path: MFMA bf16 conv tiles + fp32 accumulate"), SURVEY 8(d) "C5".

Between the 5x5 first layer and HPP every activation, saved tensor and gradient is bf16 ([pixel][c], half the bytes of fp32); the
3x3 layers run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; weight gradients, master weights and Adam stay fp32.  The
sequence of steps is engine_h2.py's (one launch per layer for all modalities); there are no block exponents to keep.
"""
from __future__ import annotations

import torch

from . import bf16, ops

F32 = torch.float32
U8 = torch.uint8
I16 = torch.int16
I32 = torch.int32
NBINS, FEAT, HIDDEN = 62, 128, 256
LAYERS3 = (("a2", 32, 32, 64, True), ("b1", 32, 64, 32, False), ("b2", 64, 64, 32, True), ("a3", 32, 64, 32, False),
           ("a4", 64, 64, 32, True), ("b3", 64, 128, 16, False), ("b4", 128, 128, 16, False), ("a5", 64, 128, 16, False),
           ("a6", 128, 128, 16, False))


class BFState:
    """Packed bf16 filters (both directions) and the cache of bf16 / fp32 buffers of one modality branch."""

    def __init__(self, enc):
        self.enc = enc
        dev = enc.store.device
        self.pk = {name: tuple(torch.empty((9 * cin * cout,), dtype=I16, device=dev) for _ in (0, 1)) for name, cin, cout, _, _ in LAYERS3}
        self.pooled = {name: pool for name, _, _, _, pool in LAYERS3}
        self.bufs = {}

    def pack_jobs(self):
        return [(self.enc.W(name), self.pk[name][d], d == 1, self.pooled[name]) for name, *_ in LAYERS3 for d in (0, 1)]

    def wf(self, name):
        return self.pk[name][0]

    def wd(self, name):
        return self.pk[name][1]

    def t(self, key, shape, dtype=I16):
        buf = self.bufs.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = self.bufs[key] = torch.empty(tuple(shape), dtype=dtype, device=self.enc.store.device)
        return buf


def forward_bf(encs, xs):
    S = [e.bf for e in encs]
    geo = []
    for e, x in zip(encs, xs):
        b, l = x.shape[0], x.shape[1]
        e.shape = (b, l)
        geo.append((b, l, b * l))
    k = len(encs)
    R = range(k)
    bs, l0 = [g[0] for g in geo], geo[0][1]
    assert all(g[1] == l0 for g in geo), "the modalities of a batch share the set length"
    xfs = [x.reshape(g[2], 60, 60, e.cin) for e, x, g in zip(encs, xs, geo)]
    for s, xf in zip(S, xfs):
        s.x = xf
    a1s = [bf16.conv5x5_in_fwd(xfs[i], encs[i].W("a1"), S[i].t("a1", (geo[i][2], 64, 64, 32)),
                               sign=S[i].t("a1s", (geo[i][2], 64, 64), I32)) for i in R]
    hook = getattr(encs[0], "before_conv3", None)     # (the packed 3x3 filters: repacked beside the 5x5 layer, engine.apply_gradients)
    if hook is not None:
        hook()
    p2s = [S[i].t("p2", (geo[i][2], 32, 32, 32)) for i in R]
    i2s = [S[i].t("i2", (geo[i][2], 32, 32, 32), U8) for i in R]
    bf16.conv3x3_fwd_multi(a1s, [s.wf("a2") for s in S], 32, True, p2s, i2s)
    m1s = [S[i].t("m1", (geo[i][0], 32, 32, 32)) for i in R]
    # routing words of the three set poolings (engine_h2.forward_h2): their gradients read these instead of the l frames
    routes = lambda key, hw, c: [S[i].t(key, (geo[i][0], hw, hw, 2, c), I32) for i in R] if encs[0].cfg.set_routed and l0 <= 32 else None
    bf16.setmax_fwd_multi(p2s, bs, l0, ms=m1s, routes=routes("r1", 32, 32))

    def pair_layer(na, nb, xa, xb, cout, hw, pool, ka, kb, ia=None, ib=None):
        ho = hw // 2 if pool else hw
        outs = [S[i].t(ka, (geo[i][2], ho, ho, cout)) for i in R] + [S[i].t(kb, (geo[i][0], ho, ho, cout)) for i in R]
        idxs = None
        if pool:
            idxs = [S[i].t(ia, (geo[i][2], ho, ho, cout), U8) for i in R] + [S[i].t(ib, (geo[i][0], ho, ho, cout), U8) for i in R]
        bf16.conv3x3_fwd_multi(list(xa) + list(xb), [s.wf(na) for s in S] + [s.wf(nb) for s in S], cout, pool, outs, idxs)
        return outs[:k], outs[k:]

    a3s, b1s = pair_layer("a3", "b1", p2s, m1s, 64, 32, False, "a3", "b1")
    p4s, q2s = pair_layer("a4", "b2", a3s, b1s, 64, 32, True, "p4", "q2", "i4", "j2")
    s2s = [S[i].t("s2", (geo[i][0], 16, 16, 64)) for i in R]
    bf16.setmax_fwd_multi(p4s, bs, l0, addends=q2s, sums=s2s, routes=routes("r2", 16, 64))
    a5s, b3s = pair_layer("a5", "b3", p4s, s2s, 128, 16, False, "a5", "b3")
    a6s, b4s = pair_layer("a6", "b4", a5s, b3s, 128, 16, False, "a6", "b4")
    m3s = [S[i].t("m3", (geo[i][0], 16, 16, 128), F32) for i in R]
    s3s = [S[i].t("s3", (geo[i][0], 16, 16, 128), F32) for i in R]
    bf16.setmax_fwd_f32_multi(a6s, bs, l0, m3s, b4s, s3s, routes=routes("r3", 16, 128))
    feats = ops.hpp_fwd_multi(m3s, s3s, [S[i].t("feat", (NBINS, geo[i][0], FEAT), F32) for i in R])
    outs = ops.binfc_fwd_multi(feats, [e.W("fc") for e in encs], [S[i].t("out", (NBINS, geo[i][0], HIDDEN), F32) for i in R])
    for e, o in zip(encs, outs):
        e.act = {"out": o}
    return outs


def backward_bf(encs, douts, side):
    S = [e.bf for e in encs]
    dev = encs[0].store.device
    geo = [(e.shape[0], e.shape[1], e.shape[0] * e.shape[1]) for e in encs]
    k = len(encs)
    R = range(k)
    bs, l0 = [g[0] for g in geo], geo[0][1]
    T = lambda key: [s.bufs[key] for s in S]
    RT = lambda key: T(key) if encs[0].cfg.set_routed and l0 <= 32 else None       # routing words of a set pooling (forward_bf)
    fc_args = (T("feat"), [e.W("fc") for e in encs], douts, [e.G("fc") for e in encs],
                                    [S[i].t("dfeat", (NBINS, geo[i][0], FEAT), F32) for i in R])
    ops.binfc_bwd_multi(*fc_args, parts=2)            # dfeat: the rest of the backward pass waits for it
    with side(encs[0].store.device):
        ops.binfc_bwd_multi(*fc_args, parts=1)        # the FC weight gradients: on the second stream
    dfeats = fc_args[4]
    dm3s = [S[i].t("dm3", (geo[i][0], 16, 16, 128), F32) for i in R]
    dzb4f = [S[i].t("dzb4f", (geo[i][0], 16, 16, 128), F32) for i in R]
    bf16.hpp_bwd_b4_multi(T("m3"), T("s3"), T("b4"), dfeats, dm3s, dzb4f)
    dzb4 = bf16.convert_multi(dzb4f, [S[i].t("dzb4", (geo[i][0], 16, 16, 128)) for i in R])
    dz6 = bf16.setmax_bwd_multi(T("a6"), dm3s, bs, l0, True, [S[i].t("dz6", (geo[i][2], 16, 16, 128)) for i in R], dm_is_f32=True,
                                routes=RT("r3"))

    def wgrad(na, nb, xa, xb, dza, dzb, cout, ia=None, ib=None):
        with side(dev):
            bf16.conv3x3_wgrad_multi(list(xa) + list(xb), list(dza) + list(dzb), cout, [e.G(na) for e in encs] + [e.G(nb) for e in encs],
                                     dz_idxs=None if ia is None else list(ia) + list(ib))

    def dgrad(na, nb, dza, dzb, hw, cin, cout, ka, kb, ia=None, ib=None, acta=None, actb=None):
        outa = [S[i].t(ka, (geo[i][2], hw, hw, cin)) for i in R]
        outb = [S[i].t(kb, (geo[i][0], hw, hw, cin)) for i in R]
        bf16.conv3x3_dgrad_multi(list(dza) + list(dzb), [s.wd(na) for s in S] + [s.wd(nb) for s in S], hw, cin, cout, outa + outb,
                                 dz_idxs=None if ia is None else list(ia) + list(ib),
                                 acts=None if acta is None else list(acta) + list(actb))
        return outa, outb

    wgrad("a6", "b4", T("a5"), T("b3"), dz6, dzb4, 128)
    dz5, dzb3 = dgrad("a6", "b4", dz6, dzb4, 16, 128, 128, "dz5", "dzb3", acta=T("a5"), actb=T("b3"))
    wgrad("a5", "b3", T("p4"), T("s2"), dz5, dzb3, 128)
    raw4, ds2 = dgrad("a5", "b3", dz5, dzb3, 16, 64, 128, "g4", "ds2")
    dq2 = bf16.lrelu_bwd_multi(ds2, T("q2"), [S[i].t("dq2", (geo[i][0], 16, 16, 64)) for i in R])
    dp4 = bf16.setmax_bwd_multi(T("p4"), ds2, bs, l0, True, raw4, addends=raw4, routes=RT("r2"))          # in place over the second gradient path
    wgrad("a4", "b2", T("a3"), T("b1"), dp4, dq2, 64, T("i4"), T("j2"))
    dz3, dzb1 = dgrad("a4", "b2", dp4, dq2, 32, 64, 64, "dz3", "dzb1", T("i4"), T("j2"), T("a3"), T("b1"))
    wgrad("a3", "b1", T("p2"), T("m1"), dz3, dzb1, 64)
    raw2, dm1 = dgrad("a3", "b1", dz3, dzb1, 32, 32, 64, "g2", "dm1")
    dp2 = bf16.setmax_bwd_multi(T("p2"), dm1, bs, l0, True, raw2, addends=raw2, routes=RT("r1"))
    i2 = T("i2")
    with side(dev):
        bf16.conv3x3_wgrad_multi(T("a1"), dp2, 32, [e.G("a2") for e in encs], dz_idxs=i2)
    dz1 = [S[i].t("dz1", (geo[i][2], 64, 64, 32)) for i in R]
    bf16.conv3x3_dgrad_multi(dp2, [s.wd("a2") for s in S], 64, 32, 32, dz1, dz_idxs=i2)
    with side(dev):
        for i, e in enumerate(encs):
            bf16.conv5x5_in_wgrad(S[i].x, dz1[i], e.G("a1"), sign=S[i].bufs["a1s"])
