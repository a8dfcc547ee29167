"""Parity of the f16-matrix-pipe 3x3 kernels on H2 tensors (ugaitnet_amd/csrc/conv3x3_mm.hip, wgrad3x3_mm.hip) against the
fp64 numpy oracle, at the SAME bars as the fp32 kernels they replace (tests/test_kernels_gpu.py): the split-fp16 products
must be indistinguishable from fp32 arithmetic.  Reference call sites: nets/mj_uwyhNets_ba.py:431-462."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O

pytestmark = pytest.mark.gpu

CONV_CFGS = [  # (hw, cin, cout, pool)  == the five 3x3 shapes of the encoder
    (64, 32, 32, True), (32, 32, 64, False), (32, 64, 64, True), (16, 64, 128, False), (16, 128, 128, False)]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, ref, rtol, name=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    scale = float(np.abs(ref).max()) + 1e-300
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())
    assert err <= rtol * scale, "%s: max abs err %.3e vs scale %.3e (rtol %.1e)" % (name, err, scale, rtol)


def check_meta(t, name=""):
    """The producer's bookkeeping: amax is the largest stored magnitude (taken before the value is split into halves, so equal
    to 22 bits), and it stays below 2^15."""
    e, bits = t.meta.cpu().numpy().tolist()
    amax = float(np.array([bits], np.uint32).view(np.float32)[0])
    d = t.data.cpu().numpy().view(np.float16).astype(np.float64)
    stored = np.abs(d[:, :, :, 0, :] + d[:, :, :, 1, :]).max()
    assert abs(amax - stored) <= 2.0 ** -20 * stored, "%s: meta amax %r vs stored max %r" % (name, amax, stored)
    assert amax < 2.0 ** 15, "%s: stored max %r" % (name, amax)


@pytest.mark.parametrize("scale", [1.0, 3e-7, 4e4])
def test_h2_roundtrip(dev, scale):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3, 16, 16, 32)) * scale).astype(np.float32)
    x[0, 0, 0, :4] = [0.0, -0.0, scale * 1e-9, -scale * 3e-5]     # zeros and elements far below the tensor's scale
    t = h2.encode(T(x, dev))
    check_meta(t, "encode")
    back = h2.decode(t).cpu().numpy()
    # 22 significant bits per element down to 2^-18 of the tensor's maximum, absolute 2^-40 of it below
    amax = np.abs(x).max()
    tol = np.maximum(np.abs(x.astype(np.float64)) * 2.0 ** -21, amax * 2.0 ** -38)
    assert np.all(np.abs(back.astype(np.float64) - x) <= tol), np.abs(back.astype(np.float64) - x).max() / amax
    assert np.array_equal(t.numpy().astype(np.float32), back)
    assert abs(t.true_amax() - amax) <= 1e-6 * amax


# the f16x2 kernels against the oracle on the RAW fp32 input: 22 significant bits per element below ONE exponent per tensor put up to
# 2^-22 of the tensor's maximum on every input, the filter's split another 2^-22 of its own: 4e-6 of the output's scale bounds what a
# 288 ... 1152-term sum of such errors reaches on these inputs (the cases pass at it), against 2e-6 with the rounded input fed to the oracle
H2_RAW_BAR = 4e-6


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
@pytest.mark.parametrize("xscale", [1.0, 1e-5])
def test_mm_fwd_and_dgrad(dev, hw, cin, cout, pool, xscale):
    from ugaitnet_amd import h2
    rng = np.random.default_rng(9000 + hw + cin + cout)
    n = 9 if hw <= 32 else 5   # more items than one workgroup round for the small images: exercises the item pipeline
    x = (rng.uniform(-1, 1, (n, hw, hw, cin)) * xscale).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    xt = h2.encode(T(x, dev))
    x_h2 = xt.numpy()                      # what the kernel actually multiplies (x to 22 bits)
    act = O.leaky(O.conv2d_same(x_h2, w.astype(np.float64)))
    act_raw = O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))
    wf, mf = h2.mm_pack(T(w, dev), False)
    ho = hw // 2 if pool else hw
    out = h2.H2Tensor.empty((n, ho, ho, cout), dev)
    if pool:
        idx = torch.empty((n, ho, ho, cout), dtype=torch.uint8, device=dev)
        h2.conv3x3_fwd_mm_multi([xt], [wf], [mf], cout, True, [out], [idx])
        pref, iref = O.maxpool2x2(act)
        close(out.numpy(), pref, 2e-6, "mm fwd+pool")
        # ... and against the oracle on the RAW fp32 input: the format's own 22-bit rounding of x included
        close(out.numpy(), O.maxpool2x2(act_raw)[0], H2_RAW_BAR, "mm fwd+pool, raw input")
        idx = idx.cpu().numpy()
        win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
        srt = np.sort(win, axis=3)
        clear = (srt[:, :, :, 3, :] - srt[:, :, :, 2, :]) > 1e-4 * xscale
        assert idx.max() <= 3 and np.array_equal(idx[clear], iref[clear])
    else:
        h2.conv3x3_fwd_mm_multi([xt], [wf], [mf], cout, False, [out])
        close(out.numpy(), act, 2e-6, "mm fwd")
        close(out.numpy(), act_raw, H2_RAW_BAR, "mm fwd, raw input")
    check_meta(out, "fwd out")
    # data gradient, plain and with LeakyReLU'(act of the layer's input); pooled layers take the pooled gradient + argmax
    gscale = 1e-4 * xscale               # gradients are small numbers: the block exponent has to carry them
    act_prev = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    if pool:
        dp = (rng.normal(size=(n, hw // 2, hw // 2, cout)) * gscale).astype(np.float32)
        pidx = rng.integers(0, 4, size=dp.shape).astype(np.uint8)
        dzt = h2.encode(T(dp, dev))
        dz = O.maxpool2x2_bwd(pidx, dzt.numpy())
        idx_t = [T(pidx, dev)]
    else:
        dzf = (rng.normal(size=(n, hw, hw, cout)) * gscale).astype(np.float32)
        dzt = h2.encode(T(dzf, dev))
        dz = dzt.numpy()
        idx_t = None
    _, dx_ref = O.conv2d_same_bwd(x_h2, w.astype(np.float64), dz)
    wd, md = h2.mm_pack(T(w, dev), True)
    dx = h2.H2Tensor.empty((n, hw, hw, cin), dev)
    h2.conv3x3_dgrad_mm_multi([dzt], [wd], [md], hw, cin, cout, [dx], dz_idxs=idx_t)
    close(dx.numpy(), dx_ref, 3e-6, "mm dgrad plain")
    check_meta(dx, "dgrad out")
    at = h2.encode(T(act_prev, dev))
    dx2 = h2.H2Tensor.empty((n, hw, hw, cin), dev)
    h2.conv3x3_dgrad_mm_multi([dzt], [wd], [md], hw, cin, cout, [dx2], dz_idxs=idx_t, acts=[at])
    close(dx2.numpy(), np.where(at.numpy() > 0, dx_ref, 0.3 * dx_ref), 3e-6, "mm dgrad * LeakyReLU'")


def test_mm_multi_job(dev):
    """Six jobs of one shape in one launch (three modalities x frame-level + set-level), each with its own filters, sizes and
    magnitudes: every job must equal its single-job launch bit for bit, and the oracle within the fp32 bar."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(77)
    hw, cin, cout = 16, 64, 128
    ns = [7, 5, 6, 2, 1, 3]
    scales = [1.0, 0.01, 30.0, 1.0, 1e-3, 5.0]
    xs, ws, outs, refs = [], [], [], []
    for n, s in zip(ns, scales):
        x = (rng.uniform(-1, 1, (n, hw, hw, cin)) * s).astype(np.float32)
        w = rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32)
        xs.append(h2.encode(T(x, dev)))
        ws.append(h2.mm_pack(T(w, dev), False))
        outs.append(h2.H2Tensor.empty((n, hw, hw, cout), dev))
        refs.append(O.leaky(O.conv2d_same(xs[-1].numpy(), w.astype(np.float64))))
    h2.conv3x3_fwd_mm_multi(xs, [w[0] for w in ws], [w[1] for w in ws], cout, False, outs)
    for j, (o, r) in enumerate(zip(outs, refs)):
        close(o.numpy(), r, 2e-6, "job %d" % j)
        check_meta(o, "job %d" % j)
        single = h2.H2Tensor.empty(o.shape, dev)
        h2.conv3x3_fwd_mm_multi([xs[j]], [ws[j][0]], [ws[j][1]], cout, False, [single])
        assert torch.equal(single.data, o.data) and torch.equal(single.meta, o.meta), "job %d differs from its own launch" % j


@pytest.mark.parametrize("kind", ["flat", "diagonal"])
def test_mm_pool_ties_route_to_the_first_maximum(dev, kind):
    """Exact ties of a pooling window -- identical input patches under all four outputs -- must pick position 0 like TF's
    MaxPoolGrad.  A direct convolution adds the same products in the same order for identical patches, so this holds for
    axis-aligned flats AND for 45-degree edges (where the Winograd kernels route 1.7 % of the ties elsewhere)."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(3)
    hw, cin, cout, n = 32, 64, 64, 4
    if kind == "flat":
        x = np.broadcast_to(rng.uniform(-1, 1, (n, 1, 1, cin)), (n, hw, hw, cin)).astype(np.float32)
    else:   # value depends on y - x only: outputs (0,0) and (1,1) of every window see identical patches
        yy, xx = np.meshgrid(np.arange(hw), np.arange(hw), indexing="ij")
        prof = rng.uniform(-1, 1, (n, 2 * hw, cin)).astype(np.float32)
        x = prof[:, (yy - xx) + hw, :]
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    xt = h2.encode(T(np.ascontiguousarray(x), dev))
    wf, mf = h2.mm_pack(T(w, dev), False)
    out = h2.H2Tensor.empty((n, hw // 2, hw // 2, cout), dev)
    idx = torch.empty((n, hw // 2, hw // 2, cout), dtype=torch.uint8, device=dev)
    h2.conv3x3_fwd_mm_multi([xt], [wf], [mf], cout, True, [out], [idx])
    act = O.leaky(O.conv2d_same(xt.numpy(), w.astype(np.float64)))
    _, iref = O.maxpool2x2(act)
    idx = idx.cpu().numpy()
    inner = (slice(None), slice(1, hw // 2 - 1), slice(1, hw // 2 - 1))    # windows away from the zero padding
    if kind == "flat":
        assert np.all(idx[inner] == 0)
    else:
        win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
        tied = np.abs(win[..., 0, :] - win[..., 3, :]) <= 1e-12 * np.abs(win).max()
        top = np.maximum(win[..., 0, :], win[..., 3, :]) >= np.maximum(win[..., 1, :], win[..., 2, :]) + 1e-6
        sel = np.zeros_like(tied)
        sel[inner] = (tied & top)[inner]
        assert sel.sum() > 1000
        assert np.all(idx[sel] == 0), "%d of %d diagonal ties not routed to position 0" % ((idx[sel] != 0).sum(), sel.sum())


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_mm_wgrad(dev, hw, cin, cout, pool):
    """dW = sum in (x) dz on the f16 pipe (transposed LDS reads, K = pixels) against the fp64 oracle at the fp32 kernels' bar;
    pooled layers take the pooled gradient + argmax.  Two launches must agree bit for bit (fixed-order slab reduction)."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(4000 + hw + cin + cout)
    n = 7 if hw <= 32 else 3
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    xt = h2.encode(T(x, dev))
    ho = hw // 2 if pool else hw
    g = (rng.normal(size=(n, ho, ho, cout)) * 1e-5).astype(np.float32)
    gt = h2.encode(T(g, dev))
    if pool:
        pidx = rng.integers(0, 4, size=g.shape).astype(np.uint8)
        dz = O.maxpool2x2_bwd(pidx, gt.numpy())
        idx_t = [T(pidx, dev)]
    else:
        dz = gt.numpy()
        idx_t = None
    w0 = np.zeros((3, 3, cin, cout))
    dw_ref, _ = O.conv2d_same_bwd(xt.numpy(), w0, dz, need_dx=False)
    dw = torch.empty((3, 3, cin, cout), device=dev)
    h2.conv3x3_wgrad_mm_multi([xt], [gt], cout, [dw], dz_idxs=idx_t)
    close(dw, dw_ref, 5e-6, "mm wgrad")
    dw2 = torch.empty_like(dw)
    h2.conv3x3_wgrad_mm_multi([xt], [gt], cout, [dw2], dz_idxs=idx_t)
    assert torch.equal(dw, dw2)


def test_mm_wgrad_multi_job(dev):
    """Jobs of very different sizes in one launch (frame-level + set-level of three modalities, down to ONE image): shares cross
    job boundaries, some groups get no strip at all; every job must equal its own single-job launch within rounding of the
    slab order and the oracle within the bar."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(11)
    hw, cin, cout = 16, 64, 128
    ns = [9, 5, 7, 1, 2, 1]
    xs, gs, refs = [], [], []
    for n in ns:
        xs.append(h2.encode(T(rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32), dev)))
        gs.append(h2.encode(T((rng.normal(size=(n, hw, hw, cout)) * rng.choice([1e-6, 1e-2, 3.0])).astype(np.float32), dev)))
        refs.append(O.conv2d_same_bwd(xs[-1].numpy(), np.zeros((3, 3, cin, cout)), gs[-1].numpy(), need_dx=False)[0])
    dws = [torch.empty((3, 3, cin, cout), device=dev) for _ in ns]
    h2.conv3x3_wgrad_mm_multi(xs, gs, cout, dws)
    for j, (d, r) in enumerate(zip(dws, refs)):
        close(d, r, 5e-6, "wgrad job %d" % j)
    # a launch with fewer strips than groups (one image of two strips): the unwritten slabs must count as zero
    one = torch.empty((3, 3, cin, cout), device=dev)
    h2.conv3x3_wgrad_mm_multi([xs[3]], [gs[3]], cout, [one])
    close(one, refs[3], 5e-6, "wgrad single image")


def test_mm_results_do_not_depend_on_the_persistent_grid(dev):
    """ugn_set_persistent_wgs(n < 256) leaves CUs free for RCCL's channels under data parallelism; forward and data gradient are
    bit-identical whatever the number of persistent workgroups."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(2)
    hw, cin, cout = 32, 64, 64
    ns = [11, 3]
    xs = [h2.encode(T(rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32), dev)) for n in ns]
    ws = [h2.mm_pack(T(rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32), dev), False) for _ in ns]
    res = {}
    try:
        for wgs in (0, 200, 24):
            h2.set_persistent_wgs(wgs)
            outs = [h2.H2Tensor.empty((n, hw // 2, hw // 2, cout), dev) for n in ns]
            idxs = [torch.empty((n, hw // 2, hw // 2, cout), dtype=torch.uint8, device=dev) for n in ns]
            h2.conv3x3_fwd_mm_multi(xs, [w[0] for w in ws], [w[1] for w in ws], cout, True, outs, idxs)
            res[wgs] = [(o.data.clone(), o.meta.clone(), i.clone()) for o, i in zip(outs, idxs)]
    finally:
        h2.set_persistent_wgs(0)
    for wgs in (200, 24):
        for a, b in zip(res[0], res[wgs]):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), wgs
    with pytest.raises(ValueError):
        h2.set_persistent_wgs(300)


def test_weight_gradients_on_a_reduced_persistent_grid(dev):
    """ugn_set_persistent_wgs also sizes the weight-gradient launches (what runs beside the bucketed all-reduce).
    The launch's shares and slabs are fixed by the job sizes; on a reduced grid a workgroup walks several shares in turn: the
    gradient is bit-identical on every grid."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(8)
    for hw, cin, cout, pool in [(32, 64, 64, True), (16, 128, 128, False), (64, 32, 32, True)]:
        ns = [5, 2]
        ho = hw // 2 if pool else hw
        xs = [h2.encode(T(rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32), dev)) for n in ns]
        gs = [h2.encode(T((rng.normal(size=(n, ho, ho, cout)) * 1e-3).astype(np.float32), dev)) for n in ns]
        idx = [T(rng.integers(0, 4, size=(n, ho, ho, cout)).astype(np.uint8), dev) for n in ns] if pool else None
        refs = []
        for j, n in enumerate(ns):
            dz = O.maxpool2x2_bwd(idx[j].cpu().numpy(), gs[j].numpy()) if pool else gs[j].numpy()
            refs.append(O.conv2d_same_bwd(xs[j].numpy(), np.zeros((3, 3, cin, cout)), dz, need_dx=False)[0])
        res = {}
        try:
            for wgs in (0, 224, 64):
                h2.set_persistent_wgs(wgs)
                dws = [torch.empty((3, 3, cin, cout), device=dev) for _ in ns]
                h2.conv3x3_wgrad_mm_multi(xs, gs, cout, dws, dz_idxs=idx)
                again = [torch.empty((3, 3, cin, cout), device=dev) for _ in ns]
                h2.conv3x3_wgrad_mm_multi(xs, gs, cout, again, dz_idxs=idx)
                assert all(torch.equal(a, b) for a, b in zip(dws, again)), wgs
                res[wgs] = [d.cpu().numpy() for d in dws]
        finally:
            h2.set_persistent_wgs(0)
        for wgs in (0, 224, 64):
            for d, r in zip(res[wgs], refs):
                close(d, r, 5e-6, "wgrad on %d workgroups" % wgs)
        for wgs in (224, 64):
            for a, b in zip(res[0], res[wgs]):
                assert np.array_equal(a, b), wgs


# ---- the format's weak spot as a SPEC: ONE exponent per tensor (csrc/mm_common.h:12-20) ----------------------
# An element keeps 22 significant bits down to 2^-18 of the tensor's bound; below that the error is ABSOLUTE: 2^-40 of the bound.  The
# bound a producer uses is rigorous but loose (max|in| * L1 of the filter: ~3-5 bits above the true maximum of a convolution), so for
# an IMAGE whose largest magnitude is r times the tensor's largest (r <= 1) the error relative to that image's own scale is
#       err_image / scale_image  <=  A + F / r          A = the arithmetic bar of the kernel (2e-6 ... 5e-6),  F = 2^-33
# i.e. fp32-class for images down to ~2^-14 of the largest image of the tensor, degrading in proportion below (the reference's fp32 has
# an exponent per ELEMENT and no such floor).  The tests below pin A and F per image against the fp64 oracle on the TRUE fp32 inputs
# (encode error included), with per-image magnitudes spanning 2^24.
RANGE_F = 2.0 ** -33


def _per_image(got, ref):
    """(scale, error / scale) per image"""
    n = ref.shape[0]
    sc = np.abs(ref.reshape(n, -1)).max(axis=1)
    er = np.abs(np.asarray(got, np.float64).reshape(n, -1) - ref.reshape(n, -1)).max(axis=1)
    return sc, er / np.maximum(sc, 1e-300)


def test_h2_dynamic_range_inside_one_tensor_forward(dev):
    """Forward 3x3 layer on a tensor whose IMAGES differ in magnitude by up to 2^24 (one exponent for all of them)."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(41)
    hw, cin, cout = 16, 64, 128
    ratios = np.array([1.0, 2.0 ** -4, 2.0 ** -8, 1e-4, 2.0 ** -14, 2.0 ** -16, 2.0 ** -20, 2.0 ** -24])
    n = len(ratios)
    x = (rng.uniform(-1, 1, (n, hw, hw, cin)) * ratios[:, None, None, None]).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32)
    ref = O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))        # the TRUE inputs, not their 22-bit encodings
    xt = h2.encode(T(x, dev))
    wf, mf = h2.mm_pack(T(w, dev), False)
    out = h2.H2Tensor.empty((n, hw, hw, cout), dev)
    h2.conv3x3_fwd_mm_multi([xt], [wf], [mf], cout, False, [out])
    sc, rel = _per_image(out.numpy(), ref)
    r = sc / sc.max()
    print("H2 forward, images at ratios %s of the largest: error / image scale %s" % (["%.1e" % v for v in r], ["%.1e" % v for v in rel]))
    assert np.all(rel <= 3e-6 + RANGE_F / r), (r, rel)
    assert np.all(rel[r >= 2.0 ** -14.5] <= 5e-6), (r, rel)        # the fp32-class range: 2^14 inside one tensor
    assert rel[-1] > 1e-5        # ... and the floor is real: at 2^-24 of the tensor's scale an image is NOT fp32-class


def test_h2_dynamic_range_inside_one_tensor_gradients(dev):
    """Data gradient and weight gradient with a gradient tensor whose images span 2^23 (one clip with a 1e4 x cotangent, the rest
    at 1e-3 x): per image for the data gradient; the weight gradient SUMS the images, so its error is measured against the
    contribution of the small images alone (their contribution is below the rounding of the large one either way)."""
    from ugaitnet_amd import h2
    rng = np.random.default_rng(42)
    hw, cin, cout = 16, 64, 128
    n = 8
    gsc = np.full(n, 1e-3)
    gsc[2] = 1e4
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32)
    g = (rng.normal(size=(n, hw, hw, cout)) * gsc[:, None, None, None]).astype(np.float32)
    dw_ref, dx_ref = O.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), g.astype(np.float64))
    xt, gt = h2.encode(T(x, dev)), h2.encode(T(g, dev))
    wd, md = h2.mm_pack(T(w, dev), True)
    dx = h2.H2Tensor.empty((n, hw, hw, cin), dev)
    h2.conv3x3_dgrad_mm_multi([gt], [wd], [md], hw, cin, cout, [dx])
    sc, rel = _per_image(dx.numpy(), dx_ref)
    r = sc / sc.max()
    print("H2 data gradient, images at ratios %s: error / image scale %s" % (["%.1e" % v for v in r], ["%.1e" % v for v in rel]))
    assert np.all(rel <= 4e-6 + RANGE_F / r), (r, rel)
    assert rel[2] <= 4e-6
    # weight gradient of the SMALL images alone, computed from the same tensors: the large image's slot zeroed in the oracle AND on the device
    g_small = g.copy()
    g_small[2] = 0
    dw_small_ref, _ = O.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), g_small.astype(np.float64), need_dx=False)
    dw = torch.empty((3, 3, cin, cout), device=dev)
    h2.conv3x3_wgrad_mm_multi([xt], [gt], cout, [dw])
    close(dw, dw_ref, 5e-6, "wgrad, 2^23 span")                       # the sum is dominated by the large image: fp32-class
    # what the small images contribute is resolved to  F / r  of THEIR scale (r = 1e-7): visible, not fp32-class -- the documented floor
    err_small = np.abs(dw.cpu().numpy().astype(np.float64) - dw_ref).max() / np.abs(dw_small_ref).max()
    print("H2 weight gradient: error of the sum relative to the small images' own contribution: %.2e" % err_small)
