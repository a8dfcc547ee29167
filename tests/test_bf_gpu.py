"""bf16 path (BASELINE.json configs[4]): kernels on bf16 tensors in HBM against the numpy oracle evaluated on the SAME bf16 values.
Inputs and filters are exactly representable, products are exact in the fp32 accumulator, so what remains is the rounding of the
bf16 OUTPUT (2^-9 relative) -- or fp32 accumulation order for the fp32 weight gradients.  Reference: nets/mj_uwyhNets_ba.py:428-481."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O

pytestmark = pytest.mark.gpu

CONV_CFGS = [(64, 32, 32, True), (32, 32, 64, False), (32, 64, 64, True), (16, 64, 128, False), (16, 128, 128, False)]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def bfr(a):
    """round to bf16 (nearest even) and back, in numpy"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def close(got, ref, rtol, name=""):
    scale = float(np.abs(ref).max()) + 1e-300
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    assert err <= rtol * scale, "%s: max abs err %.3e vs scale %.3e (rtol %.1e)" % (name, err, scale, rtol)


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_bf16_conv_fwd_dgrad_wgrad(dev, hw, cin, cout, pool):
    from ugaitnet_amd import bf16
    rng = np.random.default_rng(600 + hw + cin + cout)
    n = 9 if hw <= 32 else 5
    x = bfr(rng.uniform(-1, 1, (n, hw, hw, cin)))
    w = bfr(rng.uniform(-0.2, 0.2, (3, 3, cin, cout)))
    xt = bf16.from_f32(T(x.astype(np.float32), dev))
    assert np.array_equal(bf16.to_numpy(xt), x)
    wt = T(w.astype(np.float32), dev)
    act = O.leaky(O.conv2d_same(x, w))
    ho = hw // 2 if pool else hw
    out = bf16.empty((n, ho, ho, cout), dev)
    if pool:
        idx = torch.empty((n, ho, ho, cout), dtype=torch.uint8, device=dev)
        bf16.conv3x3_fwd_multi([xt], [bf16.pack(wt, False)], cout, True, [out], [idx])
        pref, iref = O.maxpool2x2(act)
        close(bf16.to_numpy(out), pref, 2.0 ** -8, "bf16 fwd+pool")
        idx = idx.cpu().numpy()
        win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
        srt = np.sort(win, axis=3)
        clear = (srt[:, :, :, 3, :] - srt[:, :, :, 2, :]) > 1e-4
        assert idx.max() <= 3 and np.array_equal(idx[clear], iref[clear])
    else:
        bf16.conv3x3_fwd_multi([xt], [bf16.pack(wt, False)], cout, False, [out])
        close(bf16.to_numpy(out), act, 2.0 ** -8, "bf16 fwd")
    # data gradient (plain / LeakyReLU'), weight gradient; pooled layers take the pooled gradient + argmax
    if pool:
        dp = bfr(rng.normal(size=(n, hw // 2, hw // 2, cout)) * 1e-3)
        pidx = rng.integers(0, 4, size=dp.shape).astype(np.uint8)
        dzt = bf16.from_f32(T(dp.astype(np.float32), dev))
        dz = O.maxpool2x2_bwd(pidx, dp)
        idx_t = [T(pidx, dev)]
    else:
        dz = bfr(rng.normal(size=(n, hw, hw, cout)) * 1e-3)
        dzt = bf16.from_f32(T(dz.astype(np.float32), dev))
        idx_t = None
    dw_ref, dx_ref = O.conv2d_same_bwd(x, w, dz)
    wd = bf16.pack(wt, True, pooled=bool(pool))
    dx = bf16.empty((n, hw, hw, cin), dev)
    bf16.conv3x3_dgrad_multi([dzt], [wd], hw, cin, cout, [dx], dz_idxs=idx_t)
    close(bf16.to_numpy(dx), dx_ref, 2.0 ** -8, "bf16 dgrad")
    act_prev = bfr(rng.normal(size=(n, hw, hw, cin)))
    at = bf16.from_f32(T(act_prev.astype(np.float32), dev))
    dx2 = bf16.empty((n, hw, hw, cin), dev)
    bf16.conv3x3_dgrad_multi([dzt], [wd], hw, cin, cout, [dx2], dz_idxs=idx_t, acts=[at])
    close(bf16.to_numpy(dx2), np.where(act_prev > 0, dx_ref, 0.3 * dx_ref), 2.0 ** -8, "bf16 dgrad * LeakyReLU'")
    dw = torch.empty((3, 3, cin, cout), device=dev)
    bf16.conv3x3_wgrad_multi([xt], [dzt], cout, [dw], dz_idxs=idx_t)
    close(dw.cpu().numpy(), dw_ref, 5e-6, "bf16 wgrad (fp32 result)")


def test_bf16_multi_job_and_first_layer(dev):
    from ugaitnet_amd import bf16
    rng = np.random.default_rng(61)
    hw, cin, cout = 16, 64, 128
    ns = [7, 5, 6, 2, 1, 3]
    xs = [bfr(rng.uniform(-1, 1, (n, hw, hw, cin))) for n in ns]
    ws = [bfr(rng.uniform(-0.1, 0.1, (3, 3, cin, cout))) for _ in ns]
    xt = [bf16.from_f32(T(x.astype(np.float32), dev)) for x in xs]
    pk = [bf16.pack(T(w.astype(np.float32), dev), False) for w in ws]
    outs = [bf16.empty((n, hw, hw, cout), dev) for n in ns]
    bf16.conv3x3_fwd_multi(xt, pk, cout, False, outs)
    for j in range(len(ns)):
        close(bf16.to_numpy(outs[j]), O.leaky(O.conv2d_same(xs[j], ws[j])), 2.0 ** -8, "job %d" % j)
    # first layer: fp32 input and filter, bf16 a1 (+ sign bits), weight gradient from a bf16 gradient
    n, c1 = 5, 2
    x = rng.uniform(-0.5, 0.5, (n, 60, 60, c1)).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (5, 5, c1, 32)).astype(np.float32)
    xf = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0))).astype(np.float64)
    ref = O.leaky(O.conv2d_same(xf, w.astype(np.float64)))
    a1 = bf16.empty((n, 64, 64, 32), dev)
    sign = torch.empty((n, 64, 64), dtype=torch.int32, device=dev)
    bf16.conv5x5_in_fwd(T(x, dev), T(w, dev), a1, sign=sign)
    close(bf16.to_numpy(a1), ref, 2.0 ** -8, "conv5x5 fwd bf16")
    dz = bfr(rng.normal(size=(n, 64, 64, 32)) * 1e-3)
    bits = ((sign.cpu().numpy().astype(np.uint32)[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    dw_ref, _ = O.conv2d_same_bwd(xf, w.astype(np.float64), dz * np.where(bits, 1.0, 0.3), need_dx=False)
    dw = torch.empty((5, 5, c1, 32), device=dev)
    bf16.conv5x5_in_wgrad(T(x, dev), bf16.from_f32(T(dz.astype(np.float32), dev)), dw, sign=sign)
    # (both operands of the bf16 MFMA are rounded to bf16: the patch values and gradient x LeakyReLU' -- 8 significant bits, as in every
    #  other layer of this path; the fp32-MFMA form of the round's first half met 5e-6 here)
    close(dw.cpu().numpy(), dw_ref, 2.0 ** -8, "conv5x5 wgrad bf16")


def test_bf16_set_pooling_and_friends(dev):
    from ugaitnet_amd import bf16, ops
    rng = np.random.default_rng(62)
    b, l, hw, c = 3, 25, 16, 64
    p = bfr(rng.normal(size=(b, l, hw, hw, c)))
    p[:, 3] = p[:, 0]
    q = bfr(rng.normal(size=(b, hw, hw, c)) * 3)
    pt = bf16.from_f32(T(p.reshape(b * l, hw, hw, c).astype(np.float32), dev))
    qt = bf16.from_f32(T(q.astype(np.float32), dev))
    m, s = bf16.empty((b, hw, hw, c), dev), bf16.empty((b, hw, hw, c), dev)
    bf16.setmax_fwd_multi([pt], [b], l, ms=[m], addends=[qt], sums=[s])
    assert np.array_equal(bf16.to_numpy(m), p.max(axis=1))
    assert np.array_equal(bf16.to_numpy(s), bfr(p.max(axis=1) + q))
    mf, sf = torch.empty((b, hw, hw, c), device=dev), torch.empty((b, hw, hw, c), device=dev)
    bf16.setmax_fwd_f32_multi([pt], [b], l, [mf], [qt], [sf])
    assert np.array_equal(mf.cpu().numpy(), p.max(axis=1).astype(np.float32)) and np.array_equal(sf.cpu().numpy(), (p.max(axis=1) + q).astype(np.float32))
    dm = bfr(rng.normal(size=(b, hw, hw, c)) * 1e-3)
    ad = bfr(rng.normal(size=(b * l, hw, hw, c)) * 1e-3)
    adt = bf16.from_f32(T(ad.astype(np.float32), dev))
    ref = (O.setmax_bwd(p, p.max(axis=1), dm) + ad.reshape(b, l, hw, hw, c)) * np.where(p > 0, 1.0, 0.3)
    bf16.setmax_bwd_multi([pt], [bf16.from_f32(T(dm.astype(np.float32), dev))], [b], l, True, [adt], addends=[adt])
    close(bf16.to_numpy(adt).reshape(b, l, hw, hw, c), ref, 2.0 ** -8, "setmax bwd bf16")
    # the routed form (what the engine runs): routing words from the forward pass, the gradient without the frames -- the same bits
    route = torch.empty((b, hw, hw, 2, c), dtype=torch.int32, device=dev)
    bf16.setmax_fwd_multi([pt], [b], l, ms=[m], routes=[route])
    words = route.cpu().numpy().view(np.uint32)
    bits = lambda mask: sum((mask[:, t].astype(np.uint32) << np.uint32(t)) for t in range(l))
    assert np.array_equal(words[:, :, :, 0], bits(p == p.max(axis=1, keepdims=True))) and np.array_equal(words[:, :, :, 1], bits(p > 0))
    adt2 = bf16.from_f32(T(ad.astype(np.float32), dev))
    bf16.setmax_bwd_multi(None, [bf16.from_f32(T(dm.astype(np.float32), dev))], [b], l, True, [adt2], addends=[adt2], routes=[route])
    assert torch.equal(adt2, adt), "routed gradient differs from the one that reads the frames"
    route3 = torch.zeros_like(route)
    bf16.setmax_fwd_f32_multi([pt], [b], l, [mf], None, None, routes=[route3])
    assert torch.equal(route3, route)
    out = bf16.empty((b, hw, hw, c), dev)
    g = bfr(rng.normal(size=(b, hw, hw, c)))
    bf16.lrelu_bwd_multi([bf16.from_f32(T(g.astype(np.float32), dev))], [qt], [out])
    assert np.array_equal(bf16.to_numpy(out), bfr(g * np.where(q > 0, 1.0, 0.3)))
    # HPP backward: only the sign of b4 enters
    a, s3, b4 = (rng.normal(size=(b, 16, 16, 128)).astype(np.float32) for _ in range(3))
    dfeat = rng.normal(size=(62, b, 128)).astype(np.float32)
    r1, r2 = ops.hpp_bwd(T(a, dev), T(s3, dev), T(b4, dev), T(dfeat, dev))
    d1, d2 = torch.empty_like(r1), torch.empty_like(r2)
    bf16.hpp_bwd_b4_multi([T(a, dev)], [T(s3, dev)], [bf16.from_f32(T(b4, dev))], [T(dfeat, dev)], [d1], [d2])
    assert torch.equal(d1, r1) and torch.equal(d2, r2)
