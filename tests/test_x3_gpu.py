"""Parity of the "x3" 3x3 kernels (ugaitnet_amd/csrc/conv3x3_x3.hip, wgrad3x3_x3.hip: IEEE fp32 tensors, products through the exact
three-way bf16 split on the bf16 matrix pipe) against the fp64 numpy oracle at the bars of the fp32 kernels they stand in for
(tests/test_kernels_gpu.py), on the RAW fp32 inputs -- there is no storage format in between -- and, side by side, against the
fp32-MFMA kernels of the library on the same inputs: the split products must be at least as close to fp64 as an fp32 MFMA chain.
Reference call sites: nets/mj_uwyhNets_ba.py:431-462."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O

pytestmark = pytest.mark.gpu

CONV_CFGS = [  # (hw, cin, cout, pool)  == the five 3x3 shapes of the encoder
    (64, 32, 32, True), (32, 32, 64, False), (32, 64, 64, True), (16, 64, 128, False), (16, 128, 128, False)]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def err_of(got, ref):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    scale = float(np.abs(ref).max()) + 1e-300
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) / scale


def close(got, ref, rtol, name=""):
    e = err_of(got, ref)
    assert e <= rtol, "%s: max abs err %.3e of the tensor's scale (bar %.1e)" % (name, e, rtol)
    return e


def test_x3_split_planes(dev):
    """x0 + x1 + x2 == x bit for bit, each plane is the round-to-nearest-even bf16 of what the planes before it left, over fp32's
    normal range, signed zeros and values that are already bf16."""
    from ugaitnet_amd import x3
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.uniform(-30, 30, 4096), rng.uniform(-1, 1, 4096),
                        [0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 + 2.0 ** -7 + 2.0 ** -23, 3.0e38, -1.0e-30, 0.1]]).astype(np.float32)
    planes = x3.split(T(x, dev)).cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    p = planes.view(np.float32).astype(np.float64)            # [3, n]
    assert np.array_equal((p[0] + p[1] + p[2]).astype(np.float32), x) and np.array_equal(p[0] + p[1] + p[2], x.astype(np.float64))

    def bf16_rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).astype(np.float64)
    assert np.array_equal(p[0], bf16_rne(x))
    assert np.array_equal(p[1], bf16_rne(x.astype(np.float64) - p[0]))
    assert np.array_equal(p[2], x.astype(np.float64) - p[0] - p[1])


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
@pytest.mark.parametrize("xscale", [1.0, 1e-5])
def test_x3_fwd_and_dgrad(dev, hw, cin, cout, pool, xscale):
    from ugaitnet_amd import ops, x3
    rng = np.random.default_rng(9000 + hw + cin + cout)
    n = 9 if hw <= 32 else 5   # more items than one workgroup round for the small images: exercises the item pipeline
    x = (rng.uniform(-1, 1, (n, hw, hw, cin)) * xscale).astype(np.float32)
    w = rng.uniform(-0.2, 0.2, (3, 3, cin, cout)).astype(np.float32)
    xt, wt = T(x, dev), T(w, dev)
    act = O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64)))
    wf = x3.pack(wt, False)
    ho = hw // 2 if pool else hw
    out = torch.empty((n, ho, ho, cout), device=dev)
    # the fp32-MFMA direct kernel of the library on the same inputs (UGN_WINO=0 path): the yardstick for "fp32-grade"
    ref32 = ops.conv3x3_fwd(xt, ops.pack3x3(wt), pool)
    if pool:
        idx = torch.empty((n, ho, ho, cout), dtype=torch.uint8, device=dev)
        x3.conv3x3_fwd_multi([xt], [wf], cout, True, [out], [idx])
        pref, iref = O.maxpool2x2(act)
        e = close(out, pref, 2e-6, "x3 fwd+pool")
        e32 = err_of(ref32[0], pref)
        idx = idx.cpu().numpy()
        win = act.reshape(n, hw // 2, 2, hw // 2, 2, cout).transpose(0, 1, 3, 2, 4, 5).reshape(n, hw // 2, hw // 2, 4, cout)
        srt = np.sort(win, axis=3)
        clear = (srt[:, :, :, 3, :] - srt[:, :, :, 2, :]) > 1e-4 * xscale
        assert idx.max() <= 3 and np.array_equal(idx[clear], iref[clear])
    else:
        x3.conv3x3_fwd_multi([xt], [wf], cout, False, [out])
        e = close(out, act, 2e-6, "x3 fwd")
        e32 = err_of(ref32, act)
    assert e <= 2.0 * e32 + 1e-8, "x3 forward error %.3e against the fp32-MFMA kernel's %.3e" % (e, e32)
    # data gradient, plain and with LeakyReLU'(act of the layer's input); pooled layers take the pooled gradient + argmax
    gscale = 1e-4 * xscale
    act_prev = rng.normal(size=(n, hw, hw, cin)).astype(np.float32)
    if pool:
        dp = (rng.normal(size=(n, hw // 2, hw // 2, cout)) * gscale).astype(np.float32)
        pidx = rng.integers(0, 4, size=dp.shape).astype(np.uint8)
        dzt = T(dp, dev)
        dz = O.maxpool2x2_bwd(pidx, dp.astype(np.float64))
        idx_t = [T(pidx, dev)]
    else:
        dzf = (rng.normal(size=(n, hw, hw, cout)) * gscale).astype(np.float32)
        dzt = T(dzf, dev)
        dz = dzf.astype(np.float64)
        idx_t = None
    _, dx_ref = O.conv2d_same_bwd(x.astype(np.float64), w.astype(np.float64), dz)
    wd = x3.pack(wt, True)
    dx = torch.empty((n, hw, hw, cin), device=dev)
    x3.conv3x3_dgrad_multi([dzt], [wd], hw, cin, cout, [dx], dz_idxs=idx_t)
    e = close(dx, dx_ref, 3e-6, "x3 dgrad plain")
    dx32 = ops.conv3x3_dgrad(dzt, wt, hw, dz_idx=None if idx_t is None else idx_t[0])
    e32 = err_of(dx32, dx_ref)
    assert e <= 2.0 * e32 + 1e-8, "x3 data-gradient error %.3e against the fp32-MFMA kernel's %.3e" % (e, e32)
    at = T(act_prev, dev)
    dx2 = torch.empty((n, hw, hw, cin), device=dev)
    x3.conv3x3_dgrad_multi([dzt], [wd], hw, cin, cout, [dx2], dz_idxs=idx_t, acts=[at])
    close(dx2, np.where(act_prev > 0, dx_ref, 0.3 * dx_ref), 3e-6, "x3 dgrad * LeakyReLU'")


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_x3_wgrad(dev, hw, cin, cout, pool):
    from ugaitnet_amd import ops, x3
    rng = np.random.default_rng(4100 + hw + cin + cout)
    n = 11 if hw <= 32 else 3
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    hz = hw // 2 if pool else hw
    dzf = (rng.normal(size=(n, hz, hz, cout)) * 1e-3).astype(np.float32)
    if pool:
        pidx = rng.integers(0, 4, size=dzf.shape).astype(np.uint8)
        dz = O.maxpool2x2_bwd(pidx, dzf.astype(np.float64))
        idx_t = [T(pidx, dev)]
    else:
        dz, idx_t = dzf.astype(np.float64), None
    dw_ref, _ = O.conv2d_same_bwd(x.astype(np.float64), np.zeros((3, 3, cin, cout)), dz)
    dw = torch.full((3, 3, cin, cout), float("nan"), device=dev)
    x3.conv3x3_wgrad_multi([T(x, dev)], [T(dzf, dev)], cout, [dw], dz_idxs=idx_t)
    e = close(dw, dw_ref, 3e-6, "x3 wgrad")
    dw32 = ops.conv3x3_wgrad(T(x, dev), T(dzf, dev), cout, dz_idx=None if idx_t is None else idx_t[0])
    e32 = err_of(dw32, dw_ref)
    assert e <= 3.0 * e32 + 1e-8, "x3 weight-gradient error %.3e against the fp32-MFMA kernel's %.3e" % (e, e32)
    # bitwise reproducible (fixed-order reduction, no atomics)
    dw2 = torch.empty_like(dw)
    x3.conv3x3_wgrad_multi([T(x, dev)], [T(dzf, dev)], cout, [dw2], dz_idxs=idx_t)
    assert torch.equal(dw, dw2)


def test_x3_multi_job(dev):
    """Six jobs of one shape in one launch (three modalities x frame-level + set-level), each with its own filters, sizes and
    magnitudes: every job must equal its single-job launch bit for bit, and the oracle within the fp32 bar."""
    from ugaitnet_amd import x3
    rng = np.random.default_rng(77)
    hw, cin, cout = 16, 64, 128
    ns = [7, 5, 6, 2, 1, 3]
    scales = [1.0, 0.01, 30.0, 1.0, 1e-3, 5.0]
    xs, ws, wfs, outs, refs, dzs, dws = [], [], [], [], [], [], []
    for n, s in zip(ns, scales):
        x = (rng.uniform(-1, 1, (n, hw, hw, cin)) * s).astype(np.float32)
        w = rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32)
        xs.append(T(x, dev))
        ws.append(T(w, dev))
        wfs.append(x3.pack(ws[-1], False))
        outs.append(torch.empty((n, hw, hw, cout), device=dev))
        refs.append(O.leaky(O.conv2d_same(x.astype(np.float64), w.astype(np.float64))))
        dzs.append(T((rng.normal(size=(n, hw, hw, cout)) * s * 1e-3).astype(np.float32), dev))
        dws.append(torch.empty((3, 3, cin, cout), device=dev))
    x3.conv3x3_fwd_multi(xs, wfs, cout, False, outs)
    x3.conv3x3_wgrad_multi(xs, dzs, cout, dws)
    for j, (o, r) in enumerate(zip(outs, refs)):
        close(o, r, 2e-6, "job %d" % j)
        single = torch.empty_like(o)
        x3.conv3x3_fwd_multi([xs[j]], [wfs[j]], cout, False, [single])
        assert torch.equal(single, o), "job %d differs from its single-job launch" % j
        dw_ref, _ = O.conv2d_same_bwd(xs[j].cpu().numpy().astype(np.float64), np.zeros((3, 3, cin, cout)),
                                      dzs[j].cpu().numpy().astype(np.float64))
        close(dws[j], dw_ref, 3e-6, "wgrad job %d" % j)


@pytest.mark.parametrize("hw,cin,cout,pool", CONV_CFGS)
def test_six_products_against_all_nine(dev, hw, cin, cout, pool):
    """The default arithmetic keeps six of the nine partial products of the split operands; `products=9` keeps all of them -- the
    EXACT product of every pair of fp32 operands, accumulated in fp32.  On the hardware, same inputs, every kernel: the two differ by
    no more than the fp32 accumulation's own rounding and have the SAME error against fp64 (measured: equal to three digits in every
    forward and data-gradient kernel, within 1 % in the weight gradients), i.e. the three dropped products are not what limits the result."""
    from ugaitnet_amd import x3
    rng = np.random.default_rng(600 + hw + cin + cout)
    n = 6 if hw <= 32 else 3
    x = rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    hz = hw // 2 if pool else hw
    dzf = (rng.standard_normal((n, hz, hz, cout)) * 1e-3).astype(np.float32)
    pidx = rng.integers(0, 4, size=dzf.shape).astype(np.uint8) if pool else None
    xt, wt, dzt = T(x, dev), T(w, dev), T(dzf, dev)
    it = [T(pidx, dev)] if pool else None
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    act = O.leaky(O.conv2d_same(x64, w64))
    fref = O.maxpool2x2(act)[0] if pool else act
    dz64 = O.maxpool2x2_bwd(pidx, dzf.astype(np.float64)) if pool else dzf.astype(np.float64)
    dw_ref, dx_ref = O.conv2d_same_bwd(x64, w64, dz64)
    wf, wd = x3.pack(wt, False), x3.pack(wt, True)
    res = {}
    for np_ in (6, 9):
        out = torch.empty((n, hz, hz, cout), device=dev)
        idx = torch.empty((n, hz, hz, cout), dtype=torch.uint8, device=dev) if pool else None
        x3.conv3x3_fwd_multi([xt], [wf], cout, pool, [out], [idx] if pool else None, products=np_)
        dx = torch.empty((n, hw, hw, cin), device=dev)
        x3.conv3x3_dgrad_multi([dzt], [wd], hw, cin, cout, [dx], dz_idxs=it, products=np_)
        dw = torch.empty((3, 3, cin, cout), device=dev)
        x3.conv3x3_wgrad_multi([xt], [dzt], cout, [dw], dz_idxs=it, products=np_)
        res[np_] = (out, dx, dw)
    for k, (name, ref) in enumerate((("forward", fref), ("data gradient", dx_ref), ("weight gradient", dw_ref))):
        e6, e9 = err_of(res[6][k], ref), err_of(res[9][k], ref)
        d69 = err_of(res[6][k], res[9][k].cpu().numpy().astype(np.float64))
        print("%s %d->%d @%d: error against fp64 with six products %.2e, with nine %.2e; six against nine %.2e" % (name, cin, cout, hw, e6, e9, d69))
        assert e6 <= 1.15 * e9 + 2e-8, (name, e6, e9)          # dropping the three smallest products costs nothing measurable ...
        # ... and the two results are about as far apart as either is from fp64 (two results with INDEPENDENT rounding, each e from the
        # truth, lie up to 2 e apart; since round 6 the weight gradients' slabs are added segment-wise -- a third of the error of the
        # former one-chain sum, 1.4e-7 instead of 3.5e-7 at 32->64 -- and what is left is exactly such independent rounding)
        assert d69 <= 1.5 * max(e6, e9) + 2e-8, (name, d69, e6, e9)


def test_x3_results_do_not_depend_on_the_persistent_grid(dev):
    """ugn_set_persistent_wgs(n < 256) leaves CUs free for RCCL's channels under data parallelism: the x3 forward / data-gradient
    launches then run n (or 2 n) persistent workgroups over the same items -- bit-identical results for every n."""
    from ugaitnet_amd import ops, x3
    rng = np.random.default_rng(5)
    hw, cin, cout, n = 32, 64, 64, 13
    x = T(rng.uniform(-1, 1, (n, hw, hw, cin)).astype(np.float32), dev)
    w = T(rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32), dev)
    dp = T((rng.normal(size=(n, hw // 2, hw // 2, cout)) * 1e-3).astype(np.float32), dev)
    pidx = T(rng.integers(0, 4, size=(n, hw // 2, hw // 2, cout)).astype(np.uint8), dev)
    wf, wd = x3.pack(w, False), x3.pack(w, True)
    res = []
    try:
        for wgs in (0, 224, 64, 8):
            ops.set_persistent_wgs(wgs)
            out = torch.empty((n, hw // 2, hw // 2, cout), device=dev)
            idx = torch.empty((n, hw // 2, hw // 2, cout), dtype=torch.uint8, device=dev)
            x3.conv3x3_fwd_multi([x], [wf], cout, True, [out], [idx])
            dx = torch.empty((n, hw, hw, cin), device=dev)
            x3.conv3x3_dgrad_multi([dp], [wd], hw, cin, cout, [dx], dz_idxs=[pidx], acts=[x])
            res.append((out, idx, dx))
    finally:
        ops.set_persistent_wgs(0)
    for r in res[1:]:
        assert all(torch.equal(a, b) for a, b in zip(res[0], r))


def test_x3_split_is_exact_and_full_range(dev):
    """The three-way split loses nothing (x0 + x1 + x2 == x bit for bit) for 2^-110 <= |x| < 3.396e38 (test_x3_edge_semantics pins what
    happens outside) -- no block exponent, no
    dependence on the other elements of a tensor: a convolution of one image is the same inside any batch and at any scale."""
    from ugaitnet_amd import x3
    rng = np.random.default_rng(3)
    hw, cin, cout = 16, 64, 128
    w = rng.uniform(-0.1, 0.1, (3, 3, cin, cout)).astype(np.float32)
    wf = x3.pack(T(w, dev), False)
    base = rng.uniform(-1, 1, (1, hw, hw, cin)).astype(np.float32)
    ref = O.leaky(O.conv2d_same(base.astype(np.float64), w.astype(np.float64)))
    alone = torch.empty((1, hw, hw, cout), device=dev)
    x3.conv3x3_fwd_multi([T(base, dev)], [wf], cout, False, [alone])
    for k in (-80, -24, 0, 30, 90):
        s = np.float32(2.0) ** k
        # the same image scaled by a power of two, in a batch whose other images are 2^40 times larger
        batch = np.concatenate([base * s, base * s * np.float32(2.0) ** 20, base * s * np.float32(2.0) ** -20]).astype(np.float32)
        out = torch.empty((3, hw, hw, cout), device=dev)
        x3.conv3x3_fwd_multi([T(batch, dev)], [wf], cout, False, [out])
        got = out[0:1].cpu().numpy().astype(np.float64) / float(s)
        assert np.array_equal(got, alone.cpu().numpy().astype(np.float64)), "2^%d: scaling by a power of two must commute bit for bit" % k
        close(got, ref, 2e-6, "scale 2^%d" % k)


def test_a_later_core_never_inherits_an_earlier_cores_grid(dev):
    """the persistent grid is the library's one process-wide setting; every core of a persistent kernel set (x3 included,
    the default arithmetic) sets it from ITS settings at construction -- a reduced grid (Settings.persistent_wgs, or 224 by itself when
    the bucketed all-reduce overlaps the backward pass of several ranks) does not outlive the core that asked for it."""
    from ugaitnet_amd import engine, ops
    try:
        small = engine.GaitCore([1], nclasses=4, multimodal=False, conv_precision="bf16", config=engine.DEFAULTS.replace(persistent_wgs=64))
        assert small.persistent_wgs == 64 and ops.get_persistent_wgs() == 64
        x3core = engine.GaitCore([1], nclasses=4, multimodal=False, conv_precision="f32x3")
        assert x3core.persistent_wgs == 0 and ops.get_persistent_wgs() == 256
        named = engine.GaitCore([1], nclasses=4, multimodal=False, conv_precision="f32x3", config=engine.DEFAULTS.replace(persistent_wgs=128))
        assert ops.get_persistent_wgs() == 128
        # what a rank of a data-parallel job with the overlapped all-reduce gets by itself
        assert engine.GaitCore.grid_for(engine.DEFAULTS.replace(ar_overlap=True), 8) == 224
        assert engine.GaitCore.grid_for(engine.DEFAULTS.replace(ar_overlap=True), 1) == 0
        assert engine.GaitCore.grid_for(engine.DEFAULTS.replace(ar_overlap=False), 8) == 0
        del small, named
    finally:
        ops.set_persistent_wgs(0)


def test_x3_edge_semantics(dev):
    """What the x3 arithmetic does OUTSIDE the range where its split is exact, pinned beside the library's direct fp32-MFMA kernel on
    the same tensors (measured by tools/x3_edges.py, stated in INTEGRATION.md "Known differences"):
      * exact split, results bit-identical to the fp32 kernel: 2^-110 <= |x| < 3.3961775e38 (0x7f7f8000, where bf16(x) rounds to inf);
      * |x| >= 3.3961775e38, +-inf: the first plane is inf, the residual inf - inf = NaN -- the output is NaN where an fp32
        convolution gives a finite value or +-inf (and, like an fp32 convolution's inf * 0, NaN reaches the neighbouring outputs);
        NaN stays NaN;
      * |x| < 2^-110 (operands whose planes would be bf16-subnormal below 2^-133): the bits of x below 2^-133 are dropped -- an
        ABSOLUTE error of at most 2^-133 per operand (relative 2^-23 at 2^-112, 2^-8 at 2^-126; fp32 subnormals below 2^-133 become 0).
        bf16-subnormal PLANES themselves are multiplied exactly (the matrix pipe does not flush them)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import x3_edges
    rows = {r["name"]: r for r in x3_edges.run(dev)}
    for name in ("+inf", "-inf", "nan", "flt_max", "-flt_max", "tie_to_inf", "w = inf", "w = flt_max"):
        r = rows[name]
        assert np.isnan(r["x3"]), (name, r["x3"])
        want = r["want"]
        assert (np.isnan(want) and np.isnan(r["direct"])) or r["direct"] == np.float32(want), (name, r["direct"], want)
    for name in ("bf16_max", "below_tie_to_inf", "3.0e38", "1.0", "2^-100*(1+2^-23+2^-9)", "2^-105*(1+2^-23+2^-9)", "2^-110*(1+2^-23+2^-9)",
                 "tiny product 2^-70*2^-70"):
        r = rows[name]
        assert r["split_exact_x"] and r["split_exact_w"], name
        assert r["x3"] == r["direct"] == float(np.float32(r["want"])), (name, r["x3"], r["direct"], r["want"])
        assert r["x3_elsewhere_clean"] and r["direct_elsewhere_clean"], name
    tiny = 2.0 ** -133
    for name in ("2^-112*(1+2^-23+2^-9)", "2^-116*(1+2^-23+2^-9)", "2^-120*(1+2^-23+2^-9)", "2^-124*(1+2^-23+2^-9)", "2^-126*(1+2^-23+2^-9)",
                 "subnormal 2^-130", "subnormal 2^-149"):
        r = rows[name]
        assert r["direct"] == float(np.float32(r["want"])), name                  # the fp32 kernel keeps every bit, subnormals included
        assert abs(r["x3"] - r["want"]) <= tiny, (name, r["x3"], r["want"])       # x3: what lies below 2^-133 is dropped, nothing more
        assert r["x3_elsewhere_clean"], name
    assert rows["subnormal 2^-130"]["x3"] == rows["subnormal 2^-130"]["want"]      # a bf16-subnormal plane is multiplied exactly
    assert rows["2^-112*(1+2^-23+2^-9)"]["x3"] != rows["2^-112*(1+2^-23+2^-9)"]["want"]     # (the loss is real: pinned, not hidden)
