"""The steps around the 3x3 layers on H2 tensors (ugaitnet_amd/csrc/h2_elem.hip, conv5x5.hip H2 variants) against the numpy
oracle evaluated on the values the H2 inputs hold.  Reference lines: nets/mj_uwyhNets_ba.py:428-430 (first layer),
:435,451-452,463-465 (set pooling + Add), :468-481 (HPP)."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, ref, rtol, name=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    scale = float(np.abs(ref).max()) + 1e-300
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())
    assert err <= rtol * scale, "%s: max abs err %.3e vs scale %.3e (rtol %.1e)" % (name, err, scale, rtol)


def bound_ok(t, name=""):
    """meta.amax of these kernels is a BOUND of the stored magnitudes (below 2^15), not their exact maximum."""
    e, bits = t.meta.cpu().numpy().tolist()
    amax = float(np.array([bits], np.uint32).view(np.float32)[0])
    d = t.data.cpu().numpy().view(np.float16).astype(np.float64)
    stored = np.abs(d[:, :, :, 0, :] + d[:, :, :, 1, :]).max()
    assert stored <= amax * (1 + 2.0 ** -20) and amax < 2.0 ** 15, "%s: stored max %r, meta bound %r" % (name, stored, amax)


@pytest.mark.parametrize("cin", [1, 2])
@pytest.mark.parametrize("xscale", [1.0, 1e-6])
def test_conv5x5_h2(dev, cin, xscale):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(20 + cin)
    n = 5
    x = (rng.uniform(-0.5, 0.5, (n, 60, 60, cin)) * xscale).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (5, 5, cin, 32)).astype(np.float32)
    xf = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)))
    ref = O.leaky(O.conv2d_same(xf.astype(np.float64), w.astype(np.float64)))
    xm = torch.zeros(2, dtype=torch.int32, device=dev)
    xd = T(x, dev)
    h2.absmax_multi([xd], [xm])
    a1 = h2.H2Tensor.empty((n, 64, 64, 32), dev)
    sign = torch.empty((n, 64, 64), dtype=torch.int32, device=dev)
    h2.conv5x5_in_fwd_h2(xd, xm, T(w, dev), a1, sign=sign)
    close(a1.numpy(), ref, 2e-6, "conv5x5 fwd h2")
    e, bits = a1.meta.cpu().numpy().tolist()
    amax = float(np.array([bits], np.uint32).view(np.float32)[0])
    assert abs(amax * 2.0 ** -e - np.abs(a1.numpy()).max()) <= 1e-6 * np.abs(ref).max() and amax < 2.0 ** 15
    bits_ = ((sign.cpu().numpy().astype(np.uint32)[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    assert np.array_equal(bits_, a1.numpy() > 0)
    # weight gradient from an H2 gradient (small numbers: the exponent has to carry them) + the LeakyReLU' bits
    dz = (rng.normal(size=(n, 64, 64, 32)) * 1e-5).astype(np.float32)
    dzt = h2.encode(T(dz, dev))
    dz_eff = dzt.numpy() * np.where(a1.numpy() > 0, 1.0, 0.3)
    dw_ref, _ = O.conv2d_same_bwd(xf.astype(np.float64), w.astype(np.float64), dz_eff, need_dx=False)
    dw = torch.empty((5, 5, cin, 32), device=dev)
    h2.conv5x5_in_wgrad_h2(xd, dzt, dw, sign=sign)
    close(dw, dw_ref, 5e-6, "conv5x5 wgrad h2")
    # the same on the f16 matrix pipe (round 4; what the engine runs): x split into f16 halves with the exponent of max|x|, LeakyReLU' as
    # 0.3 * sum + 0.7 * (sum over the pixels with a1 > 0); and without the sign bits (plain product)
    dw2 = torch.empty((5, 5, cin, 32), device=dev)
    h2.conv5x5_in_wgrad_h2(xd, dzt, dw2, sign=sign, x_meta=xm)
    close(dw2, dw_ref, 5e-6, "conv5x5 wgrad h2 on the f16 pipe")
    again = torch.empty_like(dw2)
    h2.conv5x5_in_wgrad_h2(xd, dzt, again, sign=sign, x_meta=xm)
    assert torch.equal(dw2, again)
    dw_plain_ref, _ = O.conv2d_same_bwd(xf.astype(np.float64), w.astype(np.float64), dzt.numpy(), need_dx=False)
    dw3 = torch.empty((5, 5, cin, 32), device=dev)
    h2.conv5x5_in_wgrad_h2(xd, dzt, dw3, sign=None, x_meta=xm)
    close(dw3, dw_plain_ref, 5e-6, "conv5x5 wgrad h2 on the f16 pipe, no LeakyReLU'")


def _frames(rng, b, l, hw, c, scale, ties=True):
    p = (rng.normal(size=(b, l, hw, hw, c)) * scale).astype(np.float32)
    if ties:      # exact ties over the frame axis: the maximum of some positions appears two or three times
        p[:, 3] = p[:, 0]
        p[:, 7, :, :, ::2] = p[:, 5, :, :, ::2]
    return p


@pytest.mark.parametrize("hw,c", [(32, 32), (16, 64), (16, 128)])
def test_setmax_fwd_h2(dev, hw, c):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(hw + c)
    b, l = 3, 25
    p = _frames(rng, b, l, hw, c, 2.0)
    pt = h2.encode(T(p.reshape(b * l, hw, hw, c), dev))
    q = (rng.normal(size=(b, hw, hw, c)) * 300.0).astype(np.float32)       # another exponent than p's
    qt = h2.encode(T(q, dev))
    pv = pt.numpy().reshape(b, l, hw, hw, c)
    m_ref = pv.max(axis=1)
    m = h2.H2Tensor.empty((b, hw, hw, c), dev)
    s = h2.H2Tensor.empty((b, hw, hw, c), dev)
    h2.setmax_fwd_h2_multi([pt], [b], l, ms=[m], addends=[qt], sums=[s])
    assert np.array_equal(m.numpy(), m_ref), "the maximum of stored values is one of them: exact"
    close(s.numpy(), m_ref + qt.numpy(), 3e-7, "m + addend")
    bound_ok(m, "m"), bound_ok(s, "sum")
    # maxima only (no addend), and the fp32-output form that feeds HPP
    m2 = h2.H2Tensor.empty((b, hw, hw, c), dev)
    h2.setmax_fwd_h2_multi([pt], [b], l, ms=[m2])
    assert np.array_equal(m2.numpy(), m_ref)
    mf, sf = torch.empty((b, hw, hw, c), device=dev), torch.empty((b, hw, hw, c), device=dev)
    h2.setmax_fwd_h2_f32_multi([pt], [b], l, [mf], [qt], [sf])
    assert np.array_equal(mf.cpu().numpy(), m_ref.astype(np.float32))
    close(sf, m_ref + qt.numpy(), 2e-7, "fp32 sum")


@pytest.mark.parametrize("hw,c,with_add,f32dm", [(32, 32, True, False), (16, 64, True, False), (16, 128, False, True)])
def test_setmax_bwd_h2(dev, hw, c, with_add, f32dm):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(100 + hw + c)
    b, l = 3, 25
    p = _frames(rng, b, l, hw, c, 1.0)
    pt = h2.encode(T(p.reshape(b * l, hw, hw, c), dev))
    pv = pt.numpy().reshape(b, l, hw, hw, c)
    dm = (rng.normal(size=(b, hw, hw, c)) * 1e-5).astype(np.float32)
    if f32dm:
        dmd = T(dm, dev)
        dmm = torch.zeros(2, dtype=torch.int32, device=dev)
        h2.absmax_multi([dmd], [dmm])
        dms, dm_metas, dmv = [dmd], [dmm], dm.astype(np.float64)
    else:
        dmt = h2.encode(T(dm, dev))
        dms, dm_metas, dmv = [dmt], [dmt.meta], dmt.numpy()
    ref = O.setmax_bwd(pv, pv.max(axis=1), dmv)
    adds = None
    out = h2.H2Tensor.empty((b * l, hw, hw, c), dev)
    if with_add:
        ad = (rng.normal(size=(b * l, hw, hw, c)) * 3e-4).astype(np.float32)
        adt = h2.encode(T(ad, dev))
        ref = ref + adt.numpy().reshape(b, l, hw, hw, c)
        adds = [adt]
        out = h2.H2Tensor(adt.data, torch.zeros(2, dtype=torch.int32, device=dev))     # in place over the addend, own meta
    ref = ref * np.where(pv > 0, 1.0, 0.3)
    keep = adt.data.clone() if with_add else None
    h2.setmax_bwd_h2_multi([pt], dms, dm_metas, [b], l, True, [out], addends=adds, dm_is_f32=f32dm)
    close(out.numpy().reshape(b, l, hw, hw, c), ref, 3e-7, "setmax bwd h2")
    bound_ok(out, "setmax bwd out")
    # the routed form (what the engine runs): the forward pass writes which frames hold the maximum / are positive, the gradient
    # reads those words instead of the frames -- the same bits out, and no frame pointer at all
    route = torch.empty((b, hw, hw, 2, c), dtype=torch.int32, device=dev)
    h2.setmax_fwd_h2_multi([pt], [b], l, ms=[h2.H2Tensor.empty((b, hw, hw, c), dev)], routes=[route])
    words = route.cpu().numpy().view(np.uint32)
    bits = lambda mask: sum((mask[:, t].astype(np.uint32) << np.uint32(t)) for t in range(l))
    assert np.array_equal(words[:, :, :, 0], bits(pv == pv.max(axis=1, keepdims=True))), "maximum bits (ties: several frames)"
    assert np.array_equal(words[:, :, :, 1], bits(pv > 0)), "sign bits"
    first = out.data.clone()
    first_meta = out.meta.clone()
    out2 = h2.H2Tensor.empty((b * l, hw, hw, c), dev)
    if with_add:
        adt.data.copy_(keep)
        out2 = h2.H2Tensor(adt.data, torch.zeros(2, dtype=torch.int32, device=dev))
    h2.setmax_bwd_h2_multi(None, dms, dm_metas, [b], l, True, [out2], addends=adds, dm_is_f32=f32dm, routes=[route])
    assert torch.equal(out2.data, first) and torch.equal(out2.meta, first_meta), "routed gradient differs from the one that reads the frames"
    # fp32-output forward (the last set pooling) writes the same words
    route3 = torch.zeros_like(route)
    mf = torch.empty((b, hw, hw, c), device=dev)
    h2.setmax_fwd_h2_f32_multi([pt], [b], l, [mf], None, None, routes=[route3])
    assert torch.equal(route3, route)


def test_lrelu_bwd_and_encode_multi(dev):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(8)
    shapes = [(24, 16, 16, 64), (5, 16, 16, 64), (9, 16, 16, 64)]
    gs = [(rng.normal(size=s) * 10.0 ** rng.integers(-7, 2)).astype(np.float32) for s in shapes]
    acts = [rng.normal(size=s).astype(np.float32) for s in shapes]
    xs = [T(g, dev) for g in gs]
    scratch = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in shapes]
    h2.absmax_multi(xs, scratch)
    gt = [h2.H2Tensor.empty(s, dev) for s in shapes]
    h2.encode_multi(xs, scratch, gt)
    for g, t in zip(gs, gt):
        single = h2.encode(T(g, dev))
        assert torch.equal(single.data, t.data) and torch.equal(single.meta, t.meta)
    at = [h2.encode(T(a, dev)) for a in acts]
    outs = [h2.H2Tensor.empty(s, dev) for s in shapes]
    h2.lrelu_bwd_h2_multi(gt, at, outs)
    for g, a, o in zip(gt, at, outs):
        close(o.numpy(), g.numpy() * np.where(a.numpy() > 0, 1.0, 0.3), 3e-7, "lrelu bwd h2")
        bound_ok(o)


def test_hpp_bwd_with_h2_b4(dev):
    """Only the sign of b4 enters: the H2 form must reproduce the fp32 kernel bit for bit."""
    from ugaitnet_amd import h2, ops
    rng = np.random.default_rng(12)
    b = 5
    a, s3, b4 = (rng.normal(size=(b, 16, 16, 128)).astype(np.float32) for _ in range(3))
    b4[0, 0, 0, :7] = 0.0
    dfeat = rng.normal(size=(62, b, 128)).astype(np.float32)
    dm3_ref, dzb4_ref = ops.hpp_bwd(T(a, dev), T(s3, dev), T(b4, dev), T(dfeat, dev))
    dm3, dzb4 = torch.empty_like(dm3_ref), torch.empty_like(dzb4_ref)
    h2.hpp_bwd_b4h2_multi([T(a, dev)], [T(s3, dev)], [h2.encode(T(b4, dev))], [T(dfeat, dev)], [dm3], [dzb4])
    assert torch.equal(dm3, dm3_ref) and torch.equal(dzb4, dzb4_ref)
