"""MaxPool tie-breaking on the Winograd fp32 path (conv_precision="f32"), on the fp32 direct kernels (UGN_WINO=0) and on the H2
path (conv_precision="h2", the default: direct convolutions on the f16 matrix pipe, exact on every tie).

Reference rule: TF's MaxPoolGrad routes the gradient to the FIRST maximum of a 2x2 window in row-major order
(nets/mj_uwyhNets_ba.py:433,449).  Exact ties are the normal case on silhouettes and constant inputs: wherever the 3x3 input
patches under two positions of a window are identical, the two pre-pool activations are equal.

What the Winograd path guarantees, and why.  For a 4x4 patch that is constant along x (or y, or both) over the rows (columns)
the tied outputs depend on, the input transform B^T d B is exactly zero in every column (row) but one -- v - v = 0 in any
arithmetic -- so the products with those points are exact zeros, the output transform adds exact zeros and the tied outputs of
the tile come out bit-identical; the epilogue's strict `>` scan then picks the first one.  That holds whenever the layer's
INPUT is bit-identical at positions with identical neighbourhoods, which is true of a1 (the 5x5 layer is a direct
convolution): the first pooled layer (a2 -> i2) reproduces first-max on every axis-aligned tie, bit for bit.
The deeper pooled layers (a4 -> i4, b2 -> j2) read the output of a NON-pooled Winograd layer (a3, b1), whose four outputs of
a tile are computed by four different formulas: at the rim of a flat region two mathematically equal a3 values can differ in
the last bit, the tie of the reference is then no tie here and the larger rounding wins.  Likewise a patch constant along a
DIAGONAL (a 45-degree silhouette edge) ties outputs (0,0) and (1,1) in exact arithmetic only.  Both move the gradient between
positions whose input patches are IDENTICAL: forward values and weight gradients of that layer do not change, only the
placement of the data gradient.  The tests below count how often it happens on tie-heavy batches and bound the effect on
every parameter gradient; the direct kernels (UGN_WINO=0, `Settings.use_winograd = False`) reproduce first-max everywhere and
are checked bit-exactly on the same batches."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O

pytestmark = pytest.mark.gpu


def _rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _flat_batch(b=6, l=3, seed=3):
    """3 modalities, all flags 1 (gradients flow everywhere): modality 0 = block silhouettes in both flow channels, 1 = constant
    frames + the generator's constant 1e-9 tensor (data/...repetitions.py:102) with flag 1, 2 = 0/1 block silhouettes."""
    rng = np.random.default_rng(seed)

    def blocks(shape_c):
        x = np.zeros((b, l, 60, 60, shape_c), np.float32)
        for i in range(b):
            for t in range(l):
                for _ in range(3):
                    y0, x0 = rng.integers(0, 44, 2)
                    h, w = rng.integers(8, 28, 2)
                    x[i, t, y0:y0 + h, x0:x0 + w, :] = 1.0
        return x
    of = blocks(2) * np.float32(0.3)
    const = np.empty((b, l, 60, 60, 1), np.float32)
    const[:] = rng.uniform(-0.5, 0.5, (b, l, 1, 1, 1)).astype(np.float32)   # every frame one value
    const[0] = 1e-9                                                          # the "disabled modality" tensor, but flag 1
    sil = blocks(1)
    uses = [np.ones((b, 1), np.float32) for _ in range(3)]
    labels = np.repeat(np.arange(b // 2), 2).astype(np.int64)
    onehot = np.eye(4, dtype=np.float32)[labels]
    return [of, const, sil], uses, labels, onehot


def _diag_batch(b=6, l=3, seed=4):
    """Silhouettes bounded by 45-degree edges (both orientations) in every modality."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:60, 0:60]
    xs = []
    for cin in (2, 1, 1):
        x = np.zeros((b, l, 60, 60, cin), np.float32)
        for i in range(b):
            for t in range(l):
                k1, k2 = rng.integers(-20, 20, 2)
                m = ((xx - yy) > k1) & ((xx + yy) > 50 + k2)
                x[i, t, m, :] = 1.0
        xs.append(x * np.float32(0.3 if cin == 2 else 1.0))
    uses = [np.ones((b, 1), np.float32) for _ in range(3)]
    labels = np.repeat(np.arange(b // 2), 2).astype(np.int64)
    onehot = np.eye(4, dtype=np.float32)[labels]
    return xs, uses, labels, onehot


def _run(dev, batch, direct=False, precision="f32"):
    from ugaitnet_amd import engine
    from ugaitnet_amd.engine import GaitCore
    assert engine.DEFAULTS.use_winograd, "conv_precision='f32' means the Winograd kernels unless UGN_WINO=0"
    xs, uses, labels, onehot = batch
    rng = np.random.default_rng(21)
    p64 = dict(branches=[O.init_branch_params(rng, c, np.float64) for c in (2, 1, 1)], head=O.init_head_params(rng, 4, np.float64))
    # direct: the direct implicit-GEMM fp32 kernels (what UGN_WINO=0 selects) -- a setting of THIS core (ugaitnet_amd/config.py)
    cfg = engine.DEFAULTS.replace(use_winograd=False) if direct else None
    core = GaitCore([2, 1, 1], nclasses=4, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), device=dev,
                    conv_precision=precision, config=cfg)
    core.set_params_numpy(O.cast_params(p64, np.float32))
    core.forward_backward(xs, uses, labels, onehot)
    torch.cuda.synchronize()
    core._test_ctx = (xs, uses, labels, onehot, p64)
    r, g = O.model_loss_and_grads([x.astype(np.float64) for x in xs], [u.astype(np.float64) for u in uses], labels,
                                  onehot.astype(np.float64), p64, margin=0.2, loss_weights=(1.0, 0.1))
    return core, r, g


def _tie_windows(prepool):
    """(tied, first): windows of a pre-pool activation [n,h,w,c] whose maximum is attained more than once, and the FIRST position
    (row-major) attaining it -- the reference's routing.  "Attained" = within 1e-12 of the maximum, relatively: the fp64 oracle computes
    its convolutions with BLAS, whose blocking can round two mathematically identical sums 1e-16 apart (seen on the set-level
    maps); such windows are exact ties of the reference, which evaluates every output pixel with the same operation order."""
    n, h, w, c = prepool.shape
    xw = prepool.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
    mx = xw.max(axis=3, keepdims=True)
    near = xw >= mx - 1e-12 * np.abs(mx)
    return near.sum(axis=3) > 1, np.argmax(near, axis=3).astype(np.uint8)


def _grad_errors(core, g):
    got = core.get_grads_numpy()
    worst = {}
    for mi in range(3):
        for k, ref in g["branches"][mi].items():
            worst["m%d.%s" % (mi, k)] = _rell2(got["branches"][mi][k], ref)
    for k, ref in g["head"].items():
        worst["head." + k] = _rell2(got["head"][k], ref)
    return worst


def _tie_report(core, r, floor=0.0):
    """per pooled layer: (tied windows of the reference, how many of them the HIP path routed differently, mismatches elsewhere).
    floor > 0: only windows whose maximum magnitude reaches floor x the tensor's largest magnitude are counted (the H2 format keeps
    22 bits down to 2^-18 of a tensor's bound; the flat batch holds one clip at 1e-9 of the others' scale with its flag on)."""
    rep = {}
    for mi, enc in enumerate(core.encoders):
        c = r["branch"][mi]
        for pre, ref_idx, key in ((c["a2"], c["i2"], "i2"), (c["a4"], c["i4"], "i4"), (c["b2"], c["j2"], "j2")):
            got_idx = (enc.h2.bufs[key] if core.h2 else enc.act[key]).cpu().numpy()
            ties, first = _tie_windows(pre)
            n, h, w, ch = pre.shape
            wmax = np.abs(pre).reshape(n, h // 2, 2, w // 2, 2, ch).max(axis=(2, 4))
            keep = wmax >= floor * np.abs(pre).max()
            t, m, o = rep.get(key, (0, 0, 0))
            rep[key] = (t + int((ties & keep).sum()), m + int((got_idx[ties & keep] != first[ties & keep]).sum()),
                        o + int((got_idx[~ties & keep] != ref_idx[~ties & keep]).sum()))
    return rep


def _check_forward_and_grads(core, r, g, bar):
    """bar: relative L2 per parameter tensor against the oracle with ITS OWN routing.  A tensor beyond 1e-3 (never beyond `bar`) must
    be explained by routing: the fp64 oracle forced to the HIP path's decisions (tests/routing.py forced_step_grads; every exact
    tie of these batches then routes as the HIP path routed it) reproduces the HIP gradients to 5e-5."""
    assert abs(core.losses()["loss"] - float(r["loss"])) <= 1e-4
    assert np.abs(core.sig.cpu().numpy() - r["signature"]).max() <= 1e-3      # forward values do not depend on the routing
    worst = _grad_errors(core, g)
    assert max(worst.values()) <= bar, worst
    if max(worst.values()) > 1e-3:
        from tests import routing as R
        xs, uses, labels, onehot, p64 = core._test_ctx
        routes = [R.hip_routing(core, mi) for mi in range(3)]
        gf = R.forced_step_grads([torch.from_numpy(x.astype(np.float64)) for x in xs], [torch.from_numpy(u.astype(np.float64)) for u in uses],
                                 labels, onehot, p64, routes, core.sel.cpu().numpy())
        wf = R.grad_errors(core.get_grads_numpy(), gf)
        print("  worst tensor against the oracle's own routing %.2e (%s); forced to the HIP path's routing: worst %.2e (%s)"
              % (max(worst.values()), max(worst, key=worst.get), max(wf.values()), max(wf, key=wf.get)))
        assert max(wf.values()) <= 5e-5, wf
    return max(worst.values())


def test_flat_regions_default_path(dev):
    """Axis-aligned flat regions (block silhouettes, constant frames, the constant 1e-9 tensor with its flag on): the Winograd
    path routes EVERY tied window of all three pooled layers to the reference's first maximum."""
    core, r, g = _run(dev, _flat_batch())
    rep = _tie_report(core, r)
    print("flat-region batch, Winograd path: (ties, moved, other mismatches) per pooled layer:", rep)
    assert rep["i2"][0] > 1000000 and rep["i4"][0] > 300000 and rep["j2"][0] > 50000      # the batch really is tie-heavy
    assert all(v[1] == 0 for v in rep.values()), rep
    assert all(v[2] <= 20 for v in rep.values()), rep               # elsewhere: only fp32-vs-fp64 near-ties may differ
    _check_forward_and_grads(core, r, g, 1e-3)


def test_flat_regions_direct_kernels(dev):
    core, r, g = _run(dev, _flat_batch(), direct=True)
    rep = _tie_report(core, r)
    assert all(v[1] == 0 and v[2] <= 20 for v in rep.values()), rep
    _check_forward_and_grads(core, r, g, 1e-2)


def test_diagonal_edges_default_path_bounded(dev):
    """45-degree edges: the two diagonal outputs of a window are equal in exact arithmetic only.  Measured on this batch: 1-1.5 %
    of the tied windows route to (1,1) instead of the reference's (0,0); every parameter gradient stays within 3e-4 rel-L2."""
    core, r, g = _run(dev, _diag_batch())
    rep = _tie_report(core, r)
    ties = sum(v[0] for v in rep.values())
    moved = sum(v[1] for v in rep.values())
    worst = _check_forward_and_grads(core, r, g, 2e-3)
    print("diagonal-edge batch, Winograd path: %d of %d tied windows routed differently from first-max %r; worst "
          "parameter-gradient rel-L2 %.2e" % (moved, ties, rep, worst))
    assert moved <= 0.03 * ties
    assert all(v[2] <= 20 for v in rep.values()), rep


def test_diagonal_edges_direct_kernels_are_exact(dev):
    core, r, g = _run(dev, _diag_batch(), direct=True)
    rep = _tie_report(core, r)
    assert all(v[1] == 0 and v[2] <= 20 for v in rep.values()), rep
    _check_forward_and_grads(core, r, g, 1e-2)


@pytest.mark.parametrize("batch", ["flat", "diagonal"])
def test_h2_path_routes_every_tie_to_the_first_maximum(dev, batch):
    """conv_precision="h2" (direct convolutions on the f16 matrix pipe, ugaitnet_amd/csrc/conv3x3_mm.hip): identical input
    patches give bit-identical sums, so EVERY exact tie of the reference -- axis-aligned flats and 45-degree edges, in all three
    pooled layers -- routes to the first maximum, as TF's MaxPoolGrad does."""
    core, r, g = _run(dev, _flat_batch() if batch == "flat" else _diag_batch(), precision="h2")
    # (windows more than 2^-17 below the tensor's maximum are outside the format's full-precision range: the flat batch's one clip
    #  of constant 1e-9 with its flag on lives there -- in the reference's fp32 too it contributes 1e-9 of the others)
    rep = _tie_report(core, r, floor=2.0 ** -17)
    ties = sum(v[0] for v in rep.values())
    print("%s batch, H2 path: (ties, moved, other mismatches) per pooled layer: %r; without the magnitude floor: %r"
          % (batch, rep, _tie_report(core, r)))
    assert ties > 100000
    assert all(v[1] == 0 for v in rep.values()), rep
    assert all(v[2] <= 20 for v in rep.values()), rep               # elsewhere: only fp32-vs-fp64 near-ties may differ
    # (every such near-tie routes one element of a gradient elsewhere than the fp64 oracle does: with the handful the structured
    #  batches produce -- 0 ... 6 per layer, which ones depends on the summation order of the kernels -- single tensors of the
    #  8-clip set-level branch differ by up to 8e-3 in relative L2; forward values and losses are held to 1e-3 / 1e-4 above)
    _check_forward_and_grads(core, r, g, 1e-2)
