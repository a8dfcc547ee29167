"""MaxPool tie-breaking on the DEFAULT path (Winograd F(2x2,3x3) convolutions).

Reference rule: TF's MaxPoolGrad routes the gradient to the FIRST maximum of a 2x2 window in row-major order
(nets/mj_uwyhNets_ba.py:433,449).  Exact ties are the normal case on silhouettes and constant inputs: wherever the four 3x3
input patches under a window are identical, the four pre-pool activations are equal.

Why Winograd keeps those ties bit-exact where they come from FLAT input: for a 4x4 patch that is constant along x (or y, or
both) the input transform B^T d B is exactly zero in every column (row) but one -- v - v = 0 in any arithmetic -- so the
products with those points are exact zeros, the output transform adds exact zeros and the tied outputs of the tile come out
bit-identical; the epilogue's strict `>` scan then picks the first one.  test_flat_regions_* pin that on the whole engine.
What it does NOT cover: a patch that is constant along a DIAGONAL (a 45-degree edge crossing all 8x8 pixels under the window
in every channel).  There outputs (0,0) and (1,1) are equal in exact arithmetic, bit-equal in a direct convolution, but take
different rounding paths through the Winograd transforms.  test_diagonal_edges_* measures how often that moves a routing
decision and bounds its effect on every parameter gradient."""
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


def _run(dev, batch):
    from ugaitnet_amd import engine
    from ugaitnet_amd.engine import GaitCore
    assert engine.USE_WINOGRAD, "this test is about the default (Winograd) path"
    xs, uses, labels, onehot = batch
    rng = np.random.default_rng(21)
    p64 = dict(branches=[O.init_branch_params(rng, c, np.float64) for c in (2, 1, 1)], head=O.init_head_params(rng, 4, np.float64))
    core = GaitCore([2, 1, 1], nclasses=4, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), device=dev)
    core.set_params_numpy(O.cast_params(p64, np.float32))
    core.forward_backward(xs, uses, labels, onehot)
    torch.cuda.synchronize()
    r, g = O.model_loss_and_grads([x.astype(np.float64) for x in xs], [u.astype(np.float64) for u in uses], labels,
                                  onehot.astype(np.float64), p64, margin=0.2, loss_weights=(1.0, 0.1))
    return core, r, g


def _tie_windows(prepool):
    """Windows of a pre-pool activation [n,h,w,c] whose maximum occurs more than once (exact ties in the fp64 oracle)."""
    n, h, w, c = prepool.shape
    xw = prepool.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
    return (xw == xw.max(axis=3, keepdims=True)).sum(axis=3) > 1


def _grad_errors(core, g):
    got = core.get_grads_numpy()
    worst = {}
    for mi in range(3):
        for k, ref in g["branches"][mi].items():
            worst["m%d.%s" % (mi, k)] = _rell2(got["branches"][mi][k], ref)
    for k, ref in g["head"].items():
        worst["head." + k] = _rell2(got["head"][k], ref)
    return worst


def test_flat_regions_first_max_and_gradients(dev):
    core, r, g = _run(dev, _flat_batch())
    total_ties = 0
    for mi, enc in enumerate(core.encoders):
        c = r["branch"][mi]
        for pre, ref_idx, key in ((c["a2"], c["i2"], "i2"), (c["a4"], c["i4"], "i4"), (c["b2"], c["j2"], "j2")):
            got_idx = enc.act[key].cpu().numpy()
            ties = _tie_windows(pre)
            total_ties += int(ties.sum())
            # every exactly tied window routes to the reference's first maximum, bit for bit
            assert np.array_equal(got_idx[ties], ref_idx[ties]), (mi, key, int((got_idx[ties] != ref_idx[ties]).sum()), int(ties.sum()))
            # elsewhere only fp32-vs-fp64 near-ties may differ
            assert (got_idx != ref_idx).mean() <= 1e-4, (mi, key, float((got_idx != ref_idx).mean()))
    assert total_ties > 100000      # the batch really is tie-heavy
    assert abs(core.losses()["loss"] - float(r["loss"])) <= 1e-4
    assert np.abs(core.sig.cpu().numpy() - r["signature"]).max() <= 1e-3
    worst = _grad_errors(core, g)
    assert max(worst.values()) <= 5e-3, worst


def test_diagonal_edges_bounded_effect(dev):
    """45-degree edges: ties between the two diagonal outputs of a window are equal in exact arithmetic only; the Winograd
    path may route them to (1,1) instead of the reference's (0,0).  Guard: it stays rare and moves no gradient by more than the
    bar the near-tie flips of any fp32 implementation already need."""
    core, r, g = _run(dev, _diag_batch())
    moved, ties_all = 0, 0
    for mi, enc in enumerate(core.encoders):
        c = r["branch"][mi]
        for pre, ref_idx, key in ((c["a2"], c["i2"], "i2"), (c["a4"], c["i4"], "i4"), (c["b2"], c["j2"], "j2")):
            got_idx = enc.act[key].cpu().numpy()
            ties = _tie_windows(pre)
            moved += int((got_idx[ties] != ref_idx[ties]).sum())
            ties_all += int(ties.sum())
    print("diagonal-edge batch: %d of %d exactly tied windows routed differently from first-max" % (moved, ties_all))
    assert moved <= 0.02 * ties_all
    assert abs(core.losses()["loss"] - float(r["loss"])) <= 1e-4
    assert np.abs(core.sig.cpu().numpy() - r["signature"]).max() <= 1e-3      # forward values do not depend on the routing
    worst = _grad_errors(core, g)
    assert max(worst.values()) <= 2e-2, worst
