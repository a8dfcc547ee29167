"""Per-kernel parity: every C-ABI entry point against the numpy oracle on identical seeded inputs (GPU only)."""
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
    scale = float(np.abs(ref).max()) + 1e-30
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())
    assert err <= rtol * scale, "%s: max abs err %.3e vs scale %.3e (rtol %.1e)" % (name, err, scale, rtol)


@pytest.mark.parametrize("x3", [False, True], ids=["f32mfma", "x3"])
@pytest.mark.parametrize("cin", [1, 2])
def test_conv5x5_in_fwd_and_wgrad(dev, cin, x3):
    """x3: the form the default fp32-tensor set runs (three-way bf16 split, six products: csrc/x3_common.h) -- the same bars as the
    fp32-MFMA form, on more tiles than the persistent grids have workgroups (1,120 against 1,024 / 512: the tile loops run)."""
    import functools
    from ugaitnet_amd import ops as ops_
    ops = type("O5", (), dict(conv5x5_in_fwd=staticmethod(functools.partial(ops_.conv5x5_in_fwd, x3=x3)),
                              conv5x5_in_wgrad=staticmethod(functools.partial(ops_.conv5x5_in_wgrad, x3=x3))))
    rng = np.random.default_rng(10 + cin)
    n = 70 if x3 else 5
    x = rng.uniform(-0.5, 0.5, (n, 60, 60, cin)).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (5, 5, cin, 32)).astype(np.float32)
    xf = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)))
    ref = O.leaky(O.conv2d_same(xf.astype(np.float64), w.astype(np.float64)))
    got = ops.conv5x5_in_fwd(T(x, dev), T(w, dev))
    close(got, ref, 2e-6, "conv5x5 fwd")
    dz = rng.normal(size=(n, 64, 64, 32)).astype(np.float32)
    dw_ref, _ = O.conv2d_same_bwd(xf.astype(np.float64), w.astype(np.float64), dz.astype(np.float64), need_dx=False)
    dw = ops.conv5x5_in_wgrad(T(x, dev), T(dz, dev))
    close(dw, dw_ref, 5e-6, "conv5x5 wgrad")
    # LeakyReLU' of a1 from sign bits: fwd emits one bit per element, wgrad(dL/da1, bits) == wgrad(dL/da1 * slope(a1))
    sign = torch.empty((n, 64, 64), dtype=torch.int32, device=dev)
    got2 = ops.conv5x5_in_fwd(T(x, dev), T(w, dev), sign=sign)
    assert torch.equal(got2, got)
    bits = ((sign.cpu().numpy().astype(np.uint32)[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    assert np.array_equal(bits, got.cpu().numpy() > 0)
    dz_scaled = dz * np.where(got.cpu().numpy() > 0, 1.0, 0.3).astype(np.float32)
    dw_a = ops.conv5x5_in_wgrad(T(x, dev), T(dz, dev), sign=sign)
    dw_b = ops.conv5x5_in_wgrad(T(x, dev), T(dz_scaled, dev))
    assert torch.equal(dw_a, dw_b)


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_conv3x3_fwd(dev, hw, cin, cout, pool):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(hw + cin + cout)
    n = 3
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    act = O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))
    wp = ops.pack3x3(T(w, dev))
    assert np.array_equal(wp.cpu().numpy(), w.reshape(9, cin, cout).transpose(0, 2, 1))
    if not pool:
        close(ops.conv3x3_fwd(T(x, dev), wp, False), act, 2e-6, "conv3x3 fwd")
        return
    out, idx = ops.conv3x3_fwd(T(x, dev), wp, True)
    pref, iref = O.maxpool2x2(act)
    close(out, pref, 2e-6, "conv3x3+pool fwd")
    # the index must point at a (near-)maximum of the oracle's window; it must equal the oracle's where the max is clear
    idx = idx.cpu().numpy()
    assert idx.max() <= 3
    win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
    picked = np.take_along_axis(win, idx[:, :, :, None, :].astype(np.int64), axis=3)[:, :, :, 0, :]
    assert np.abs(picked - pref).max() <= 1e-5
    srt = np.sort(win, axis=3)
    clear = (srt[:, :, :, 3, :] - srt[:, :, :, 2, :]) > 1e-4
    assert np.array_equal(idx[clear], iref[clear])


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_conv3x3_winograd_fwd_and_dgrad(dev, hw, cin, cout, pool):
    """Winograd F(2x2,3x3) path: same operator contract, fp32 rounding-level agreement with the fp64 oracle."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(7000 + hw + cin + cout)
    n = 9 if hw <= 32 else 5   # more items than one workgroup round for the small images: exercises the item pipeline
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    act = O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))
    uf = ops.wino_pack(T(w, dev), False)
    if pool:
        out, idx = ops.conv3x3_fwd_wino(T(x, dev), uf, cout, True)
        pref, iref = O.maxpool2x2(act)
        close(out, pref, 4e-6, "wino fwd+pool")
        idx = idx.cpu().numpy()
        win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
        srt = np.sort(win, axis=3)
        clear = (srt[:, :, :, 3, :] - srt[:, :, :, 2, :]) > 1e-4
        assert idx.max() <= 3 and np.array_equal(idx[clear], iref[clear])
    else:
        close(ops.conv3x3_fwd_wino(T(x, dev), uf, cout, False), act, 4e-6, "wino fwd")
    # data gradient (with and without the fused epilogue; pooled layers get their gradient at pooled resolution + argmax map)
    act_prev = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    addend = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    if pool:
        dp = rng.normal(size=(n, hw // 2, hw // 2, cout)).astype(np.float32)
        pidx = rng.integers(0, 4, size=dp.shape).astype(np.uint8)
        dz = O.maxpool2x2_bwd(pidx, dp)
        dz_t, idx_t = T(dp, dev), T(pidx, dev)
    else:
        dz = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
        dz_t, idx_t = T(dz, dev), None
    _, dx_ref = O.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), dz.astype(np.float64))
    ud = ops.wino_pack(T(w, dev), True, pooled_dz=bool(pool))
    close(ops.conv3x3_dgrad_wino(dz_t, ud, hw, cin, cout, dz_idx=idx_t), dx_ref, 5e-6, "wino dgrad plain")
    raw = torch.empty((n, hw, hw, cin), device=dev)
    got = ops.conv3x3_dgrad_wino(dz_t, ud, hw, cin, cout, dz_idx=idx_t, act=T(act_prev, dev), addend=T(addend, dev), raw_out=raw)
    t = dx_ref + addend
    close(raw, t, 5e-6, "wino dgrad raw")
    close(got, np.where(act_prev > 0, t, 0.3 * t), 5e-6, "wino dgrad fused")


def rel_l2(got, ref):
    g = got.detach().cpu().numpy().astype(np.float64) if hasattr(got, "detach") else np.asarray(got, np.float64)
    return float(np.linalg.norm(g - ref) / np.linalg.norm(ref))


def test_wino_pack_multi_matches_single(dev):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(77)
    ws = [T(rng.normal(size=(3, 3, ci, co)).astype(np.float32), dev) for ci, co in ((32, 32), (64, 64), (64, 128), (128, 128))]
    jobs, refs = [], []
    for w in ws:
        for dg, pooled in ((False, False), (True, False), (True, True)):   # wide and narrow filter layouts
            jobs.append((w, torch.zeros(16 * w.shape[2] * w.shape[3], device=dev), dg, pooled))
            refs.append(ops.wino_pack(w, dg, pooled_dz=pooled))
    ops.wino_pack_multi(jobs)
    for (_, u, _, _), r in zip(jobs, refs):
        assert torch.equal(u, r)
    # the packed tensor is a permutation of G g G^T: same multiset of values in every layout
    a = ops.wino_pack(ws[1], True, pooled_dz=False).sort().values
    b = ops.wino_pack(ws[1], True, pooled_dz=True).sort().values
    assert torch.equal(a, b)


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS[1:])
def test_winograd_pair_launch_equals_two_single_launches(dev, hw, cin, cout, pool):
    """The frame-level layer and its set-level twin as two jobs of one launch: bit-identical to separate launches."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(4000 + hw + cin + cout)
    ns = (7, 2)
    ho = hw // 2 if pool else hw
    xs = [T(rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32), dev) for n in ns]
    ws = [T(rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32), dev) for _ in ns]
    dzs = [T(rng.normal(size=(n, ho, ho, cout)).astype(np.float32), dev) for n in ns]
    idxs = [T(rng.integers(0, 4, size=(n, ho, ho, cout)).astype(np.uint8), dev) for n in ns] if pool else None
    acts = [T(rng.normal(size=(n, hw, hw, cin)).astype(np.float32), dev) for n in ns]
    ufs = [ops.wino_pack(w, False) for w in ws]
    uds = [ops.wino_pack(w, True, pooled_dz=bool(pool)) for w in ws]
    # forward
    outs = [torch.empty((n, ho, ho, cout), device=dev) for n in ns]
    oidx = [torch.empty((n, ho, ho, cout), device=dev, dtype=torch.uint8) for n in ns] if pool else None
    ops.conv3x3_fwd_wino_pair(xs, ufs, cout, pool, outs, oidx)
    for k in range(2):
        ref = ops.conv3x3_fwd_wino(xs[k], ufs[k], cout, pool)
        if pool:
            assert torch.equal(outs[k], ref[0]) and torch.equal(oidx[k], ref[1])
        else:
            assert torch.equal(outs[k], ref)
    # data gradient with the LeakyReLU' epilogue
    gouts = [torch.empty((n, hw, hw, cin), device=dev) for n in ns]
    ops.conv3x3_dgrad_wino_pair(dzs, uds, hw, cin, cout, gouts, dz_idxs=idxs, acts=acts)
    for k in range(2):
        ref = ops.conv3x3_dgrad_wino(dzs[k], uds[k], hw, cin, cout, dz_idx=None if idxs is None else idxs[k], act=acts[k])
        assert torch.equal(gouts[k], ref)
    # weight gradient
    dws = [torch.empty((3, 3, cin, cout), device=dev) for _ in ns]
    ops.conv3x3_wgrad_wino_pair(xs, dzs, cout, dws, dz_idxs=idxs)
    for k in range(2):
        ref = ops.conv3x3_wgrad_wino(xs[k], dzs[k], cout, dz_idx=None if idxs is None else idxs[k])
        # (the split of the persistent workgroups between the jobs changes the summation order)
        close(dws[k], ref.cpu().numpy(), 2e-6, "pair wgrad job %d" % k)
    with pytest.raises(ValueError):   # mixed epilogues are rejected
        ops.conv3x3_dgrad_wino_pair(dzs, uds, hw, cin, cout, gouts, dz_idxs=idxs, acts=[acts[0], None])


@pytest.mark.parametrize("hw,cin,cout", [(32, 32, 64), (16, 64, 128)])
def test_routed_data_gradient_equals_setmax_bwd_plus_addend(dev, hw, cin, cout):
    """(dgrad + set-max gradient of the layer's output) * LeakyReLU': formed in the epilogue vs materialised by setmax_bwd."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(60 + hw)
    b, l = 3, 5
    n = b * l
    p = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    p[:, :4] = 0.0                                   # blank rows: every frame ties at the maximum there
    p[1::2, 10] = p[0::2, 10][: p[1::2, 10].shape[0]]   # and some two-way ties
    dm = rng.normal(size=(b, hw, hw, cin)).astype(np.float32)
    dz = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    ud = ops.wino_pack(T(w, dev), True)
    p_t, dm_t, dz_t = T(p, dev), T(dm, dev), T(dz, dev)
    m, cnt = ops.setmax_fwd_cnt(p_t, b, l)
    m_ref = ops.setmax_fwd(p_t, b, l)
    assert torch.equal(m, m_ref)
    cnt_ref = (p.reshape(b, l, hw, hw, cin) == p.reshape(b, l, hw, hw, cin).max(1, keepdims=True)).sum(1)
    assert np.array_equal(cnt.cpu().numpy(), cnt_ref.astype(np.float32))
    g = ops.setmax_bwd(p_t, dm_t, b, l, False)
    ref = ops.conv3x3_dgrad_wino(dz_t, ud, hw, cin, cout, act=p_t, addend=g)
    got = ops.conv3x3_dgrad_wino_routed(dz_t, ud, hw, cin, cout, p_t, m, ops.div(dm_t, cnt), l)
    assert torch.equal(got, ref)                     # same arithmetic: dm / cnt, added before the LeakyReLU' factor


def test_conv3x3_pool_first_max_on_ties(dev):
    """Constant input -> every window is a 4-way tie in the interior: the FIRST element (index 0) must win."""
    from ugaitnet_amd import ops
    x = np.full((1, 64, 64, 32), 0.25, np.float32)
    w = np.full((3, 3, 32, 32), 0.01, np.float32)
    out, idx = ops.conv3x3_fwd(T(x, dev), ops.pack3x3(T(w, dev)), True)
    idx = idx.cpu().numpy()
    act = O.leaky(O.conv2d_same(x, w))
    _, iref = O.maxpool2x2(act)
    assert np.array_equal(idx[:, 1:-1, 1:-1], iref[:, 1:-1, 1:-1])
    assert (idx[:, 1:-1, 1:-1] == 0).all()


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_conv3x3_dgrad_and_wgrad(dev, hw, cin, cout, pool):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(1000 + hw + cin + cout)
    n = 3
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    act_prev = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    addend = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    if pool:
        dp = rng.normal(size=(n, hw // 2, hw // 2, cout)).astype(np.float32)
        idx = rng.integers(0, 4, size=dp.shape).astype(np.uint8)
        dz = O.maxpool2x2_bwd(idx, dp)
        dz_t, idx_t = T(dp, dev), T(idx, dev)
    else:
        dz = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
        dz_t, idx_t = T(dz, dev), None
    dw_ref, dx_ref = O.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), dz.astype(np.float64))
    # plain data gradient
    got = ops.conv3x3_dgrad(dz_t, T(w, dev), hw, dz_idx=idx_t)
    close(got, dx_ref, 3e-6, "dgrad plain")
    # fused epilogue: (t + addend) * lrelu'(act), raw copy
    raw = torch.empty_like(got)
    got2 = ops.conv3x3_dgrad(dz_t, T(w, dev), hw, dz_idx=idx_t, act=T(act_prev, dev), addend=T(addend, dev), raw_out=raw)
    t = dx_ref + addend
    close(raw, t, 3e-6, "dgrad raw")
    close(got2, np.where(act_prev > 0, t, 0.3 * t), 3e-6, "dgrad fused")
    dw = ops.conv3x3_wgrad(T(x, dev), dz_t, cout, dz_idx=idx_t)
    close(dw, dw_ref, 5e-6, "wgrad")
    dww = ops.conv3x3_wgrad_wino(T(x, dev), dz_t, cout, dz_idx=idx_t)
    close(dww, dw_ref, 1e-5, "wgrad winograd")


def test_conv3x3_wgrad_many_frames(dev):
    """More tiles than persistent workgroups: exercises the tile loop and the two-level slab reduction."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(5)
    n, hw, cin, cout = 40, 64, 32, 32
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    dz = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
    dp, idx = O.maxpool2x2(dz)
    dzu = O.maxpool2x2_bwd(idx, dp)
    dw_ref, _ = O.conv2d_same_bwd(x.astype(np.float64), np.zeros((3, 3, cin, cout)), dzu.astype(np.float64), need_dx=False)
    dw = ops.conv3x3_wgrad(T(x, dev), T(dp, dev), cout, dz_idx=T(idx, dev))
    close(dw, dw_ref, 1e-5, "wgrad many frames")
    dww = ops.conv3x3_wgrad_wino(T(x, dev), T(dp, dev), cout, dz_idx=T(idx, dev))
    close(dww, dw_ref, 2e-5, "wgrad winograd many frames")
    again = ops.conv3x3_wgrad_wino(T(x, dev), T(dp, dev), cout, dz_idx=T(idx, dev))
    assert torch.equal(dww, again), "winograd wgrad must be bitwise reproducible"


@pytest.mark.parametrize("n", [1, 7, 70])
def test_conv3x3_wgrad_wino_region_loop(dev, n):
    """Odd frame counts: fewer regions than workgroups (n=1), ragged shares of the persistent loop (n=7, 70)."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(50 + n)
    hw, cin, cout = 16, 128, 128
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    dz = rng.normal(size=(n, hw, hw, cout)).astype(np.float32)
    dw_ref, _ = O.conv2d_same_bwd(x.astype(np.float64), np.zeros((3, 3, cin, cout)), dz.astype(np.float64), need_dx=False)
    dww = ops.conv3x3_wgrad_wino(T(x, dev), T(dz, dev), cout)
    close(dww, dw_ref, 2e-5, "wgrad winograd n=%d" % n)


def test_unsupported_shape_raises(dev):
    from ugaitnet_amd import ops
    x = torch.zeros((1, 8, 8, 16), device=dev)
    wp = torch.zeros((9, 16, 16), device=dev)
    with pytest.raises(ValueError):
        ops.conv3x3_fwd(x, wp, False)


@pytest.mark.parametrize("with_add", [False, True])
def test_setmax(dev, with_add):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(3)
    b, l, shape = 3, 25, (16, 16, 64)
    p = rng.normal(size=(b * l,) + shape).astype(np.float32)
    p5 = p.reshape((b, l) + shape)
    p5[:, 3] = p5[:, 7]          # exact ties between frames 3 and 7
    p5[0, :, 0, 0, :] = 0.5      # a 25-way tie
    p = p5.reshape(p.shape)
    add = rng.normal(size=(b,) + shape).astype(np.float32)
    m_ref = O.setmax(p5)
    if with_add:
        m, s = ops.setmax_fwd(T(p, dev), b, l, addend=T(add, dev))
        assert np.array_equal(s.cpu().numpy(), m_ref + add)
    else:
        m = ops.setmax_fwd(T(p, dev), b, l)
    assert np.array_equal(m.cpu().numpy(), m_ref)
    dm = rng.normal(size=(b,) + shape).astype(np.float32)
    ref = O.setmax_bwd(p5, m_ref, dm).reshape(p.shape)
    got = ops.setmax_bwd(T(p, dev), T(dm, dev), b, l, False)
    close(got, ref, 1e-6, "setmax bwd")
    got = ops.setmax_bwd(T(p, dev), T(dm, dev), b, l, True)
    close(got, O.leaky_bwd_from_out(p, ref), 1e-6, "setmax bwd + lrelu'")
    # second gradient path added before the LeakyReLU' factor, in place over the addend
    extra = rng.normal(size=p.shape).astype(np.float32)
    buf = T(extra, dev)
    got = ops.setmax_bwd(T(p, dev), T(dm, dev), b, l, True, out=buf, addend=buf)
    assert got.data_ptr() == buf.data_ptr()
    close(got, O.leaky_bwd_from_out(p, ref + extra), 1e-6, "setmax bwd + addend + lrelu'")


@pytest.mark.parametrize("l", [25, 1, 32, 7])
def test_setmax_routed(dev, l):
    """Set pooling with routing words (csrc/pool_set.hip, round 5): the words against numpy, the maxima and the gradient (plain, with
    LeakyReLU', with an in-place addend) bit for bit against the kernels that read the frames again, and against the oracle."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(30 + l)
    bs, shape = [3, 2], (16, 16, 64)
    s = int(np.prod(shape))
    ps, adds, dms = [], [], []
    for b in bs:
        p5 = rng.normal(size=(b, l) + shape).astype(np.float32)
        if l > 7:
            p5[:, 3] = p5[:, 7]          # exact ties between frames 3 and 7
        p5[0, :, 0, 0, :] = 0.5          # an l-way tie
        p5[0, :, 0, 1, :] = -0.25        # ... and a negative one
        p5[0, :, 1, 0, :] = 0.0          # zeros: not positive (LeakyReLU' = 0.3 at 0)
        ps.append(p5.reshape((b * l,) + shape))
        adds.append(rng.normal(size=(b,) + shape).astype(np.float32))
        dms.append(rng.normal(size=(b,) + shape).astype(np.float32))
    pt = [T(p, dev) for p in ps]
    ms = [torch.empty((b,) + shape, device=dev) for b in bs]
    sums = [torch.empty((b,) + shape, device=dev) for b in bs]
    routes = [torch.empty((b, s // 4, 2, 4), dtype=torch.int32, device=dev) for b in bs]
    ops.setmax_fwd_routed_multi(pt, bs, l, ms, routes, addends=[T(a, dev) for a in adds], sum_outs=sums)
    m0 = [torch.empty((b,) + shape, device=dev) for b in bs]
    ops.setmax_fwd_multi(pt, bs, l, m0)
    for j, b in enumerate(bs):
        p5 = ps[j].reshape((b, l, s))
        m_ref = p5.max(axis=1)
        assert torch.equal(ms[j], m0[j]) and np.array_equal(ms[j].cpu().numpy().reshape(b, s), m_ref)
        assert np.array_equal(sums[j].cpu().numpy().reshape(b, s), m_ref + adds[j].reshape(b, s))
        w = routes[j].cpu().numpy().view(np.uint32)                   # [b, s/4, 2, 4] -> [b, s] per word kind
        bits = (1 << np.arange(l, dtype=np.uint64))[None, :, None]
        mask_ref = ((p5 == m_ref[:, None, :]) * bits).sum(axis=1).astype(np.uint32)
        pos_ref = ((p5 > 0) * bits).sum(axis=1).astype(np.uint32)
        assert np.array_equal(w[:, :, 0, :].reshape(b, s), mask_ref) and np.array_equal(w[:, :, 1, :].reshape(b, s), pos_ref)
    dmt = [T(d, dev) for d in dms]
    for lrelu in (False, True):
        a = [torch.empty_like(p) for p in pt]
        ops.setmax_bwd_multi(pt, dmt, bs, l, lrelu, a)
        r = [torch.empty_like(p) for p in pt]
        ops.setmax_bwd_routed_multi(routes, dmt, bs, l, lrelu, r)
        assert all(torch.equal(x, y) for x, y in zip(a, r)), "routed gradient differs (lrelu=%r)" % lrelu
    extra = [rng.normal(size=p.shape).astype(np.float32) for p in ps]
    a, r = [T(e, dev) for e in extra], [T(e, dev) for e in extra]
    ops.setmax_bwd_multi(pt, dmt, bs, l, True, a, addends=a)
    ops.setmax_bwd_routed_multi(routes, dmt, bs, l, True, r, addends=r)                  # in place over the addend
    assert all(torch.equal(x, y) for x, y in zip(a, r))
    for j, b in enumerate(bs):
        p5 = ps[j].reshape((b, l) + shape)
        ref = O.setmax_bwd(p5, O.setmax(p5), dms[j]).reshape(ps[j].shape)
        close(r[j], O.leaky_bwd_from_out(ps[j], ref + extra[j]), 1e-6, "routed setmax bwd + addend + lrelu'")


def test_hpp(dev):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(4)
    b = 5
    a = rng.normal(size=(b, 16, 16, 128)).astype(np.float32)
    s3 = rng.normal(size=(b, 16, 16, 128)).astype(np.float32)
    a[0, 0, :, :] = a[0, 0, 0:1, :]   # ties inside a strip
    s3[1] = 0.125                      # a fully constant map: 256-way ties at every level
    b4 = rng.normal(size=a.shape).astype(np.float32)
    feat = ops.hpp_fwd(T(a, dev), T(s3, dev))
    close(feat, O.hpp(a.astype(np.float64), s3.astype(np.float64)), 2e-6, "hpp fwd")
    dfeat = rng.normal(size=(62, b, 128)).astype(np.float32)
    da, ds = O.hpp_bwd(a.astype(np.float64), s3.astype(np.float64), dfeat.astype(np.float64))
    dm3, dzb4 = ops.hpp_bwd(T(a, dev), T(s3, dev), T(b4, dev), T(dfeat, dev))
    close(dm3, da + ds, 2e-6, "hpp bwd dm3")
    close(dzb4, np.where(b4 > 0, ds, 0.3 * ds), 2e-6, "hpp bwd dzb4")


@pytest.mark.parametrize("b", [5, 24, 40])
def test_binfc(dev, b):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(6)
    feat = rng.normal(size=(62, b, 128)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (62, 128, 256)).astype(np.float32)
    out = ops.binfc_fwd(T(feat, dev), T(w, dev))
    close(out, O.binfc(feat.astype(np.float64), w.astype(np.float64)), 2e-6, "binfc fwd")
    dout = rng.normal(size=(62, b, 256)).astype(np.float32)
    dw_ref, df_ref = O.binfc_bwd(feat.astype(np.float64), w.astype(np.float64), dout.astype(np.float64))
    dw, df = ops.binfc_bwd(T(feat, dev), T(w, dev), T(dout, dev))
    close(dw, dw_ref, 3e-6, "binfc dW")
    close(df, df_ref, 3e-6, "binfc dfeat")


@pytest.mark.parametrize("mode", ["sign_max", "max", "avg"])
@pytest.mark.parametrize("nmod", [2, 3])
def test_gate_fuse(dev, mode, nmod):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(7)
    b = 7
    outs = [rng.normal(size=(62, b, 256)).astype(np.float32) for _ in range(nmod)]
    outs[1][:, :, :64] = -outs[0][:, :, :64]          # |.| ties with opposite sign: first index must win
    uses = [(rng.uniform(size=(b, 1)) > 0.4).astype(np.float32) for _ in range(nmod)]
    uses[0][0] = 0; uses[1][0] = 0                     # all-masked row (2-mod) -> zeros, index 0
    gs = [O.gate(o, u) for o, u in zip(outs, uses)]
    f_ref, sel_ref = O.fuse(gs, mode)
    ut = [T(u.reshape(-1), dev) for u in uses]
    fused, sel = ops.gate_fuse_fwd([T(o, dev) for o in outs], ut, mode)
    assert np.array_equal(fused.cpu().numpy(), f_ref) if mode != "avg" else True
    close(fused, f_ref, 1e-6, "fuse fwd")
    if mode != "avg":
        assert np.array_equal(sel.cpu().numpy().astype(np.int64), sel_ref)
    df = rng.normal(size=f_ref.shape).astype(np.float32)
    ref = [O.gate(d, u) for d, u in zip(O.fuse_bwd(sel_ref, df, nmod, mode), uses)]
    got = ops.gate_fuse_bwd(T(df, dev), sel, ut, mode)
    for g, r in zip(got, ref):
        close(g, r, 1e-6, "fuse bwd")


def test_l2norm_batch(dev):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(8)
    b = 24
    f = rng.normal(size=(62, b, 256)).astype(np.float32)
    f[:, :, 5] = 0.0          # an all-zero column: hits the 1e-12 clamp
    sig = ops.l2norm_batch_fwd(T(f, dev))
    y_ref, inv = O.l2norm_batch(f.astype(np.float64))
    close(sig, y_ref, 2e-6, "l2norm fwd")
    dy = rng.normal(size=f.shape).astype(np.float32)
    ref = O.l2norm_batch_bwd(f.astype(np.float64), y_ref, inv, dy.astype(np.float64))
    got = ops.l2norm_batch_bwd(T(f, dev), sig, T(dy, dev))
    # the clamped column has gradient dy * 1e6: compare per column relative
    close(got[:, :, :5], ref[:, :, :5], 3e-6, "l2norm bwd")
    close(got[:, :, 5], ref[:, :, 5], 3e-6, "l2norm bwd clamped column")


@pytest.mark.parametrize("mode", ["sign_max", "max", "avg"])
@pytest.mark.parametrize("b", [24, 5, 32])
def test_gate_and_normalisation_in_one_launch(dev, mode, b):
    """ugn_gate_norm_fwd / _bwd (what the engine runs for a local batch of at most 32 clips) against the two launches each replaces:
    the same bits in every output, forward and backward, ties and an all-masked clip included."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(70 + b)
    nmod = 3
    outs = [rng.normal(size=(62, b, 256)).astype(np.float32) for _ in range(nmod)]
    outs[1][:, :, :64] = -outs[0][:, :, :64]
    outs[2][:, :, 7] = 0.0
    outs[0][:, :, 7] = 0.0
    outs[1][:, :, 7] = 0.0           # an all-zero column: the 1e-12 clamp of the normalisation
    uses = [(rng.uniform(size=(b,)) > 0.4).astype(np.float32) for _ in range(nmod)]
    for u in uses:
        u[0] = 0                      # an all-masked clip
    ot, ut = [T(o, dev) for o in outs], [T(u, dev) for u in uses]
    fused, sel = ops.gate_fuse_fwd(ot, ut, mode)
    sig = ops.l2norm_batch_fwd(fused)
    f2, s2, g2 = torch.empty_like(fused), torch.empty_like(sel), torch.empty_like(sig)
    ops.gate_norm_fwd(ot, ut, mode, f2, s2, g2)
    assert torch.equal(f2, fused) and torch.equal(s2, sel) and torch.equal(g2, sig)
    dsig = T(rng.normal(size=(62, b, 256)).astype(np.float32), dev)
    ref = ops.gate_fuse_bwd(ops.l2norm_batch_bwd(fused, sig, dsig), sel, ut, mode)
    got = ops.gate_norm_bwd(fused, sig, dsig, sel, ut, mode, [torch.empty_like(fused) for _ in range(nmod)])
    for g, r in zip(got, ref):
        assert torch.equal(g, r)


@pytest.mark.parametrize("b,ncls", [(24, 150), (40, 74)])
def test_head(dev, b, ncls):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(9)
    sig = rng.normal(size=(62, b, 256)).astype(np.float32) * 0.2
    wc = rng.uniform(-0.02, 0.02, (62 * 256, ncls)).astype(np.float32)
    bc = rng.normal(size=(ncls,)).astype(np.float32) * 0.1
    lab = rng.integers(0, ncls, size=b)
    onehot = np.eye(ncls, dtype=np.float32)[lab]
    scale = 0.1 / b
    r = ops.head_fwd(T(sig, dev), T(wc, dev), T(bc, dev), T(onehot, dev), scale)
    logits, flat = O.head_logits(sig.astype(np.float64), wc.astype(np.float64), bc.astype(np.float64))
    loss, probs = O.softmax_xent(logits, onehot.astype(np.float64))
    close(r["probs"], probs, 5e-6, "probs")
    assert abs(float(r["row_loss"].mean()) - loss) < 1e-5 * max(1.0, abs(loss))
    dlog = (probs - onehot) * scale
    close(r["dlogits"], dlog, 5e-6, "dlogits")
    hit_ref = (probs.argmax(1) == lab).astype(np.float32)
    assert np.array_equal(r["hit"].cpu().numpy(), hit_ref)
    dsig0 = rng.normal(size=sig.shape).astype(np.float32)
    dsig = T(dsig0, dev)
    dwc, dbc = ops.head_bwd(T(sig, dev), T(wc, dev), r["dlogits"], dsig, True)
    close(dwc, flat.T @ dlog, 5e-6, "dwc")
    close(dbc, dlog.sum(0), 5e-6, "dbc")
    dflat = dlog @ wc.astype(np.float64).T
    close(dsig, dsig0 + dflat.reshape(b, 62, 256).transpose(1, 0, 2), 5e-6, "dsig accumulate")
    dsig2 = torch.empty_like(dsig)
    ops.head_bwd(T(sig, dev), T(wc, dev), r["dlogits"], dsig2, False)
    close(dsig2, dflat.reshape(b, 62, 256).transpose(1, 0, 2), 5e-6, "dsig write")


@pytest.mark.parametrize("labels", [np.repeat(np.arange(12), 2), np.repeat(np.arange(4), 10),
                                    np.array([0] * 10 + [1] * 10 + [2] * 4), np.repeat(np.arange(8), 16)])
def test_triplet(dev, labels):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(11)
    m = labels.shape[0]
    f = rng.normal(size=(62, m, 256)).astype(np.float32)
    sig, _ = O.l2norm_batch(f)
    sig = (sig * 0.6).astype(np.float32)   # distances around the 0.2 margin: a mix of active and inactive triplets
    hp, hn, kp, kn = ops.triplet_indices(labels)
    hp_ref, hn_ref, kp_ref, kn_ref = O.triplet_index_lists(labels)
    assert np.array_equal(hp, hp_ref) and np.array_equal(hn, hn_ref) and (kp, kn) == (kp_ref, kn_ref)  # bit-exact indices
    loss_ref, aux = O.triplet_all(labels, sig.astype(np.float64), 0.2)
    bl, bn, dsig = ops.triplet_fwd_bwd(T(sig, dev), T(hp, dev), T(hn, dev), kp, kn, 0.2, 1.0)
    # active-triplet counts: exact unless a hinge sits within rounding of 0
    h = aux["h"]
    margin_raw = 0.2 + (aux["dist"].reshape(62, -1)[:, hp].reshape(62, m, kp, 1) - aux["dist"].reshape(62, -1)[:, hn].reshape(62, m, 1, kn))
    fragile = (np.abs(margin_raw) < 1e-5).reshape(62, -1).sum(axis=1)
    assert np.all(np.abs(bn.cpu().numpy() - aux["num"]) <= fragile)
    assert abs(float(bl.mean()) - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    close(dsig, O.triplet_all_bwd(sig.astype(np.float64), aux), 2e-4, "triplet dsig")


def test_triplet_reference_example_through_the_hip_kernel(dev):
    """The only numeric example the reference holds -- nets/triplet_loss_all.py:113-118: 6 x 3 embeddings, labels 1,1,2,2,3,3, margin
    0.2 -- fed to the HIP path itself: zero-padded to 256 features (distances unchanged) and replicated to the 62 bins, through
    ugn_triplet_indices_host + ugn_triplet_fwd_bwd, against the python-loop brute force of tests/test_oracle_kat.py."""
    from ugaitnet_amd import ops
    from tests.test_oracle_kat import KAT_EMB, KAT_LAB, brute_force_triplet
    ref_loss, ref_active = brute_force_triplet(KAT_EMB, KAT_LAB, 0.2)
    sig = np.zeros((62, 6, 256), np.float32)
    sig[:, :, :3] = KAT_EMB.astype(np.float32)[None]
    hp, hn, kp, kn = ops.triplet_indices(KAT_LAB)
    hp_ref, hn_ref, kp_ref, kn_ref = O.triplet_index_lists(KAT_LAB)
    assert (kp, kn) == (kp_ref, kn_ref) == (2, 4) and np.array_equal(hp, hp_ref) and np.array_equal(hn, hn_ref)
    bl, bn, dsig = ops.triplet_fwd_bwd(T(sig, dev), T(hp, dev), T(hn, dev), kp, kn, 0.2, 1.0)
    bl, bn = bl.cpu().numpy(), bn.cpu().numpy()
    assert np.all(bn == ref_active), (bn[:4], ref_active)                  # active-triplet count of every bin: exact
    assert np.all(np.abs(bl - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss))), (bl[:4], ref_loss)
    # the gradient against the fp64 oracle on the same 62-bin tensor, and nothing leaks into the padded features
    _, aux = O.triplet_all(KAT_LAB, sig.astype(np.float64), 0.2)
    close(dsig, O.triplet_all_bwd(sig.astype(np.float64), aux), 2e-5, "triplet dsig (reference example)")
    assert float(dsig[:, :, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("labels", [np.repeat(np.arange(12), 2), np.repeat(np.arange(4), 10), np.array([0, 0, 0, 1, 2, 2, 3]),
                                    np.array([7, 7, 7, 7]), np.arange(5)])
def test_triplet_hard(dev, labels):
    """Batch-hard triplet loss per bin (compile_hard's TripletHardLoss): balanced and ragged label sets, an identity with one
    sample (no positive), a batch with one identity (no negative), all-singleton identities."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(13)
    m = labels.shape[0]
    f = rng.normal(size=(62, m, 256)).astype(np.float32)
    sig, _ = O.l2norm_batch(f)
    sig = (sig * 0.6).astype(np.float32)
    loss_ref, aux = O.triplet_hard(labels, sig.astype(np.float64), 0.2)
    bl, bn, dsig = ops.triplet_hard_fwd_bwd(T(sig, dev), T(labels.astype(np.int32), dev), 0.2, 1.0)
    fragile = (np.abs(aux["h"]) < 1e-5).sum(axis=1) + ((aux["h"] == 0).sum(axis=1) > 0) * 0   # hinges within rounding of 0
    assert np.all(np.abs(bn.cpu().numpy() - aux["num"]) <= fragile + (np.abs(aux["h"] - 0.0) < 1e-5).sum(axis=1))
    assert abs(float(bl.mean()) - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    close(dsig, O.triplet_hard_bwd(sig.astype(np.float64), aux), 2e-4, "triplet hard dsig")


def test_triplet_rejects_indivisible_labels(dev):
    from ugaitnet_amd import ops
    with pytest.raises(ValueError):
        ops.triplet_indices(np.array([0, 0, 0, 1, 1]))


def test_adam(dev):
    from ugaitnet_amd import ops
    rng = np.random.default_rng(12)
    n = 100003
    p = rng.normal(size=n).astype(np.float32); g = rng.normal(size=n).astype(np.float32)
    m = rng.normal(size=n).astype(np.float32) * 0.1; v = rng.uniform(0, 1, n).astype(np.float32) * 0.01
    pt, mt, vt = T(p, dev), T(m, dev), T(v, dev)
    t = 7
    lr_t = 1e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    ops.adam_step(pt, T(g, dev), mt, vt, lr_t)
    pr, mr, vr = p.copy(), m.copy(), v.copy()
    O.adam_step(pr, g, mr, vr, t)       # fp32 oracle: hyper-parameters rounded to fp32 as TF does
    close(pt, pr, 1e-6, "adam p"); close(mt, mr, 1e-6, "adam m"); close(vt, vr, 1e-6, "adam v")


def test_gather_and_scatter_rows(dev):
    """ugn_gather_rows / ugn_scatter_rows (the dense sub-batches of the mask-skipping step): pure row copies, bit for bit numpy's
    take / put along the clip axis of [B, L, 60, 60, C] inputs (axis 0) and of [62, B, 256] features (axis 1)."""
    from ugaitnet_amd import ops
    rng = np.random.default_rng(31)
    x = rng.normal(size=(7, 3, 60, 60, 1)).astype(np.float32)
    f = rng.normal(size=(62, 7, 256)).astype(np.float32)
    idx = np.array([5, 0, 3, 6], np.int64)
    it = T(idx, dev)
    assert np.array_equal(ops.gather_rows(T(x, dev), it, 0).cpu().numpy(), x[idx])
    assert np.array_equal(ops.gather_rows(T(f, dev), it, 1).cpu().numpy(), f[:, idx])
    dst = T(np.full_like(f, 7.0), dev)
    src = rng.normal(size=(62, 4, 256)).astype(np.float32)
    ops.scatter_rows(T(src, dev), it, 1, dst)
    want = np.full_like(f, 7.0)
    want[:, idx] = src
    assert np.array_equal(dst.cpu().numpy(), want)
    with pytest.raises(ValueError):         # rows that are not whole float4s
        ops.gather_rows(T(rng.normal(size=(4, 6)).astype(np.float32), dev), T(np.array([1], np.int64), dev), 0)
