"""Known-answer tests that pin the CPU oracle (the reference ships no tests or fixtures: SURVEY.md section 4)."""
import numpy as np
import pytest

from oracle import ugaitnet_oracle as O

# the only numeric example in the reference: nets/triplet_loss_all.py:115-116 (6 x 3 embeddings, labels 1,1,2,2,3,3)
KAT_EMB = np.array([[1.1, 1.2, 1.4], [1.09, 1.21, 1.41], [0.25, 0.45, 0.75], [0.23, 0.43, 0.7], [1.5, 2.5, 3.5],
                    [1.55, 2.75, 3.8]])
KAT_LAB = np.array([1, 1, 2, 2, 3, 3])


def brute_force_triplet(emb, labels, margin):
    """Independent python-loop statement of the batch-all loss for a BALANCED batch (one bin)."""
    m = len(labels)
    d = np.zeros((m, m))
    for i in range(m):
        for j in range(m):
            q = float(((emb[i] - emb[j]) ** 2).sum())
            d[i, j] = np.sqrt(q) if q > 0 else 0.0
    total, active = 0.0, 0
    for a in range(m):
        for p in range(m):
            if labels[p] != labels[a]:
                continue            # positives include p == a (distance 0), as in the reference's hp mask
            for n in range(m):
                if labels[n] == labels[a]:
                    continue
                h = margin + d[a, p] - d[a, n]
                if h > 0:
                    total += h
                    active += 1
    return (total / active if active else 0.0), active


def test_triplet_reference_example():
    loss, aux = O.triplet_all(KAT_LAB, KAT_EMB[None], 0.2)
    ref, active = brute_force_triplet(KAT_EMB, KAT_LAB, 0.2)
    assert abs(loss - ref) < 1e-12
    assert aux['num'][0] == active
    assert (aux['kp'], aux['kn']) == (2, 4)
    # hand-checkable part: anchors 0..3 are ~0.9-3 away from every negative except each other; only the
    # (class 1 <-> class 2... ) pairs closer than margin + d_ap stay active
    assert active == int((aux['h'] > 0).sum())


def test_triplet_two_bins_average_and_zero_bin():
    far = np.array([[0.0, 0, 0], [0.0, 0, 0.01], [10.0, 0, 0], [10.0, 0, 0.01]])   # no active triplet
    lab = np.array([0, 0, 1, 1])
    near = far.copy(); near[2:, 0] = 0.1
    l_far, a_far = O.triplet_all(lab, far[None], 0.2)
    assert l_far == 0.0 and a_far['num'][0] == 0
    both = np.stack([far, near])
    l_both, _ = O.triplet_all(lab, both, 0.2)
    l_near, _ = O.triplet_all(lab, near[None], 0.2)
    assert abs(l_both - 0.5 * l_near) < 1e-12     # mean over bins, empty bin counts as 0


def test_triplet_unbalanced_literal_reshape():
    """counts 10/10/4 (bs=24, repetitions=5): 216 positive pairs -> rows of 9, 360 negatives -> rows of 15;
    the reference's flatten + reshape([n,m,-1,1]) mixes anchors, and the restatement reproduces it literally."""
    lab = np.array([0] * 10 + [1] * 10 + [2] * 4)
    hp, hn, kp, kn = O.triplet_index_lists(lab)
    assert (kp, kn) == (9, 15) and hp.size == 216 and hn.size == 360
    assert hp[0] == 0 and hp[9] == 9 and hp[10] == 24          # row-major boolean_mask order
    rng = np.random.default_rng(0)
    emb = rng.normal(size=(2, 24, 8))
    loss, aux = O.triplet_all(lab, emb, 0.2)
    d = O.batch_dist(emb).reshape(2, -1)
    h = np.maximum(0.2 + d[:, hp].reshape(2, 24, 9, 1) - d[:, hn].reshape(2, 24, 1, 15), 0).reshape(2, -1)
    ref = np.mean([h[k].sum() / max((h[k] > 0).sum(), 1) for k in range(2)])
    assert abs(loss - ref) < 1e-12


def test_triplet_indivisible_raises():
    with pytest.raises(ValueError):
        O.triplet_index_lists(np.array([0, 0, 0, 1, 1]))


def test_batch_dist_exact_zero_and_gradient_mask():
    x = np.array([[[1.0, 2.0], [1.0, 2.0], [4.0, 6.0]]])
    d = O.batch_dist(x)
    assert d[0, 0, 1] == 0.0 and d[0, 0, 0] == 0.0 and abs(d[0, 0, 2] - 5.0) < 1e-12


def test_hpp_on_arange():
    a = np.arange(256, dtype=np.float64).reshape(1, 16, 16, 1).repeat(128, axis=3)
    b = -a
    f = O.hpp(a, b)
    assert f.shape == (62, 1, 128)
    # 1 bin: mean 127.5 + max 255 ; b: -127.5 + 0
    assert f[0, 0, 0] == 127.5 + 255 and f[1, 0, 0] == -127.5
    # 2 bins of a: positions 0..127 and 128..255
    assert f[2, 0, 0] == 63.5 + 127 and f[3, 0, 0] == 191.5 + 255
    # 16 bins: strip s covers 16s..16s+15; a part rows 30..45, b part rows 46..61
    assert f[30 + 5, 0, 0] == (5 * 16 + 7.5) + (5 * 16 + 15)
    assert f[46 + 5, 0, 0] == -(5 * 16 + 7.5) - (5 * 16)


def test_sign_max_ties_and_sign():
    g0 = np.array([[[1.0, -3.0, 2.0, 0.0]]]); g1 = np.array([[[-1.0, 3.0, -2.5, 0.0]]])
    f, sel = O.fuse([g0, g1], 'sign_max')
    assert f.tolist() == [[[1.0, -3.0, -2.5, 0.0]]] and sel.tolist() == [[[0, 0, 1, 0]]]   # ties -> first index
    f, sel = O.fuse([g0, g1], 'max')
    assert f.tolist() == [[[1.0, 3.0, 2.0, 0.0]]] and sel.tolist() == [[[0, 1, 0, 0]]]
    f, _ = O.fuse([g0, g1], 'avg')
    assert f.tolist() == [[[0.0, 0.0, -0.25, 0.0]]]


def test_maxpool_first_max_and_setmax_tie_split():
    x = np.zeros((1, 2, 2, 1)); x[0, 0, 1, 0] = 1.0; x[0, 1, 0, 0] = 1.0
    out, idx = O.maxpool2x2(x)
    assert out[0, 0, 0, 0] == 1.0 and idx[0, 0, 0, 0] == 1          # first maximum in row-major order
    dx = O.maxpool2x2_bwd(idx, np.full((1, 1, 1, 1), 2.0))
    assert dx[0, 0, 1, 0] == 2.0 and dx.sum() == 2.0
    p = np.array([[[3.0], [1.0], [3.0], [3.0]]])                     # [b=1, l=4, s=1]
    g = O.setmax_bwd(p, O.setmax(p), np.array([[6.0]]))
    assert g[0, :, 0].tolist() == [2.0, 0.0, 2.0, 2.0]               # reduce_max: equal split among ties


def test_leaky_relu_slope_at_zero():
    y = np.array([-1.0, 0.0, 2.0])
    assert np.allclose(O.leaky(y), [-0.3, 0.0, 2.0])
    assert O.leaky_bwd_from_out(y, np.ones(3)).tolist() == [0.3, 0.3, 1.0]    # features > 0 ? g : alpha*g


def test_l2norm_batch_axis_and_clamp():
    f = np.zeros((1, 3, 2)); f[0, :, 0] = [3.0, 0.0, 4.0]
    y, inv = O.l2norm_batch(f)
    assert np.allclose(y[0, :, 0], [0.6, 0.0, 0.8]) and np.allclose(y[0, :, 1], 0.0)
    assert abs(inv[0, 0, 1] - 1e6) < 1e-3                             # rsqrt(max(0, 1e-12))


def test_adam_first_step_is_lr_sign():
    p = np.array([1.0, 1.0]); g = np.array([0.5, -2.0]); m = np.zeros(2); v = np.zeros(2)
    O.adam_step(p, g, m, v, 1, lr=1e-3)
    assert np.allclose(p, [1.0 - 1e-3, 1.0 + 1e-3], atol=1e-9)


def test_glorot_fans_of_matmul_kernel():
    rng = np.random.default_rng(0)
    w = O.glorot_uniform(rng, (62, 128, 256))
    assert np.abs(w).max() <= np.sqrt(6.0 / (62 * (128 + 256)))       # receptive field = 62 for the rank-3 kernel
