"""CPU oracle: numpy restatement of UGaitNet's gaitset hot path (forward AND hand-derived backward).

TEST INFRASTRUCTURE ONLY.  Nothing under ugaitnet_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.

PARITY UNPINNED: the reference's arithmetic lives in TensorFlow 2.3 / tensorflow_addons, which are
not installed here and not vendored in /root/reference (`import tensorflow` -> ModuleNotFoundError,
SURVEY.md section 8c), and the reference ships no tests, golden vectors or fixtures for this path.
This file restates the published TF/Keras op semantics at the reference's own call sites and is
pinned only by (a) an independent torch-autograd implementation (oracle/torch_ref.py) and
(b) hand-computable known-answer tests (tests/test_oracle_*.py).

Each function cites the reference file:line it follows (paths relative to /root/reference).
All tensors are channels-last; `dtype` is float32 for the run the HIP path is compared with and
float64 for the master copy used to bound rounding error.
"""
from __future__ import annotations

import numpy as np

LEAKY_ALPHA = 0.3          # keras LeakyReLU() default, nets/mj_uwyhNets_ba.py:430
HPP_BINS = (1, 2, 4, 8, 16)  # nets/mj_uwyhNets_ba.py:470
NBINS = 62                 # 2 * sum(HPP_BINS)
FEAT = 128                 # channels of the last conv block
HIDDEN = 256               # MatMul hidden_dim, nets/mj_uwyhNets_ba.py:24


# --------------------------------------------------------------------------------------
# primitive ops (forward + backward)
# --------------------------------------------------------------------------------------
def conv2d_same(x, w):
    """Conv2D(padding='same', use_bias=False, stride 1), NHWC x HWIO (nets/mj_uwyhNets_ba.py:429).

    SAME padding at stride 1 = floor(k/2) zeros on each side.  Accumulated tap by tap so that a
    full-size batch never materialises an im2col matrix."""
    k = w.shape[0]
    p = k // 2
    n, h, ww, cin = x.shape
    cout = w.shape[3]
    xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
    if k * k * cin <= 64:      # the first layer (K = 25 or 50): one product with the patch matrix instead of 25 rank-1 / rank-2 updates
        return (_patches(xp, k, h, ww) @ w.reshape(k * k * cin, cout)).reshape(n, h, ww, cout)
    out = np.zeros((n * h * ww, cout), dtype=x.dtype)
    for dy in range(k):
        for dx in range(k):
            out += xp[:, dy:dy + h, dx:dx + ww, :].reshape(-1, cin) @ w[dy, dx]
    return out.reshape(n, h, ww, cout)


def _patches(xp, k, h, ww):
    """[n*h*w, k*k*cin] patch matrix of a padded NHWC tensor, taps in HWIO order (small cin only: 25 or 50 columns)."""
    return np.concatenate([xp[:, dy:dy + h, dx:dx + ww, :].reshape(-1, xp.shape[3]) for dy in range(k) for dx in range(k)], axis=1)


def conv2d_same_bwd(x, w, dz, need_dx=True):
    """Gradients of conv2d_same: dW (HWIO) and dx."""
    k = w.shape[0]
    p = k // 2
    n, h, ww, cin = x.shape
    cout = w.shape[3]
    xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
    dz2 = dz.reshape(-1, cout)
    if not need_dx and k * k * cin <= 64:      # the first layer's filter gradient: one product with the patch matrix
        return (_patches(xp, k, h, ww).T @ dz2).reshape(w.shape), None
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    for dy in range(k):
        for dx in range(k):
            dw[dy, dx] = xp[:, dy:dy + h, dx:dx + ww, :].reshape(-1, cin).T @ dz2
            if need_dx:
                dxp[:, dy:dy + h, dx:dx + ww, :] += (dz2 @ w[dy, dx].T).reshape(n, h, ww, cin)
    dxo = dxp[:, p:p + h, p:p + ww, :] if need_dx else None
    return dw, dxo


def leaky(x):
    """keras LeakyReLU(alpha=0.3) (nets/mj_uwyhNets_ba.py:430)."""
    return np.where(x > 0, x, x * x.dtype.type(LEAKY_ALPHA))


def leaky_bwd_from_out(y, dy):
    """TF LeakyReluGrad: features > 0 ? g : alpha*g.  sign(out) == sign(in) since alpha > 0."""
    return np.where(y > 0, dy, dy * y.dtype.type(LEAKY_ALPHA))


def maxpool2x2(x):
    """MaxPooling2D(2,2) VALID (nets/mj_uwyhNets_ba.py:433).  Returns (out, idx) where idx in 0..3
    is the FIRST maximum of the window in row-major scan order (TF CPU MaxPoolGrad routing)."""
    n, h, w, c = x.shape
    xw = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, c)
    idx = np.argmax(xw, axis=3)  # numpy argmax returns the first maximum
    out = np.take_along_axis(xw, idx[:, :, :, None, :], axis=3)[:, :, :, 0, :]
    return out, idx.astype(np.uint8)


def maxpool2x2_bwd(idx, dout):
    n, h2, w2, c = dout.shape
    dxw = np.zeros((n, h2, w2, 4, c), dtype=dout.dtype)
    np.put_along_axis(dxw, idx[:, :, :, None, :].astype(np.int64), dout[:, :, :, None, :], axis=3)
    return dxw.reshape(n, h2, w2, 2, 2, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h2 * 2, w2 * 2, c)


def setmax(x5):
    """tf.math.reduce_max(x, axis=1) over the L frames (nets/mj_uwyhNets_ba.py:435,451,463)."""
    return x5.max(axis=1)


def setmax_bwd(x5, m, dm):
    """TF reduce_max gradient: divided equally among all maxima."""
    eq = (x5 == m[:, None]).astype(x5.dtype)
    cnt = eq.sum(axis=1, keepdims=True)
    return eq * (dm[:, None] / cnt)


def hpp(a, b):
    """Horizontal pyramid pooling (nets/mj_uwyhNets_ba.py:468-481).  a, b: [B,16,16,128].
    For each bin count: a's strips then b's strips; strip = Reshape((bins,-1,c)) of the row-major
    H*W positions; value = mean + max.  Returns [62,B,128]."""
    bsz, h, w, c = a.shape
    feats = []
    for nb in HPP_BINS:
        for t in (a, b):
            r = t.reshape(bsz, nb, (h * w) // nb, c)
            feats.append(r.mean(axis=2) + r.max(axis=2))
    cat = np.concatenate(feats, axis=1)          # [B,62,128]
    return np.ascontiguousarray(cat.transpose(1, 0, 2))  # [62,B,128]


def hpp_bwd(a, b, dfeat):
    """dfeat [62,B,128] -> (da, db).  mean: g/n; max: g split equally among ties (reduce_max)."""
    bsz, h, w, c = a.shape
    da = np.zeros_like(a)
    db = np.zeros_like(b)
    d = dfeat.transpose(1, 0, 2)  # [B,62,128]
    row = 0
    for nb in HPP_BINS:
        for t, dt in ((a, da), (b, db)):
            n = (h * w) // nb
            r = t.reshape(bsz, nb, n, c)
            g = d[:, row:row + nb, :]            # [B,nb,128]
            mx = r.max(axis=2, keepdims=True)
            eq = (r == mx).astype(t.dtype)
            cnt = eq.sum(axis=2, keepdims=True)
            contrib = g[:, :, None, :] / t.dtype.type(n) + eq * (g[:, :, None, :] / cnt)
            dt += contrib.reshape(bsz, h, w, c)
            row += nb
    return da, db


def binfc(v, wfc):
    """MatMul layer: tf.matmul(x, kernel), x [62,B,128], kernel [62,128,256] (nets/mj_uwyhNets_ba.py:40)."""
    return np.einsum('kbi,kio->kbo', v, wfc)


def binfc_bwd(v, wfc, dx):
    return np.einsum('kbi,kbo->kio', v, dx), np.einsum('kbo,kio->kbi', dx, wfc)


def gate(x, use):
    """mj_tensor_times_scalar: tensor * scalar, use [B,1] broadcast over [62,B,256]
    (nets/mj_uwyhNets_ba.py:51-54)."""
    return x * use.reshape(1, -1, 1).astype(x.dtype)


def fuse(gs, mode):
    """fMerge over a list of gated branch outputs (nets/mj_uwyhNets_ba.py:814,1189).
    'sign_max': argmax of |.| over the modality axis, first index wins ties, signed value gathered
                (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:169-178).
    'max'     : keras Maximum (tf.maximum chain).  'avg': keras Average.
    Returns (fused, sel) with sel the selected modality index (int64; -1 for 'avg')."""
    st = np.stack(gs, axis=0)
    if mode == 'sign_max':
        sel = np.argmax(np.abs(st), axis=0)
        return np.take_along_axis(st, sel[None], axis=0)[0], sel
    if mode == 'max':
        sel = np.argmax(st, axis=0)  # first maximal argument receives the gradient (x >= y rule)
        return np.take_along_axis(st, sel[None], axis=0)[0], sel
    if mode == 'avg':
        return st.mean(axis=0), np.full(st.shape[1:], -1, dtype=np.int64)
    raise ValueError(mode)


def fuse_bwd(sel, df, nmod, mode):
    if mode == 'avg':
        return [df / df.dtype.type(nmod) for _ in range(nmod)]
    return [np.where(sel == m, df, df.dtype.type(0)) for m in range(nmod)]


def l2norm_batch(f):
    """tf.math.l2_normalize(x, axis=1) on [62,B,256]: axis 1 is the BATCH axis
    (nets/mj_uwyhNets_ba.py:817,1191).  y = x * rsqrt(max(sum_b x^2, 1e-12))."""
    ss = np.maximum((f * f).sum(axis=1, keepdims=True), f.dtype.type(1e-12))
    inv = 1.0 / np.sqrt(ss)
    return f * inv.astype(f.dtype), inv.astype(f.dtype)


def l2norm_batch_bwd(f, y, inv, dy):
    """SURVEY Appendix A.2.  Where the max() clamps (sum <= 1e-12) the norm is a constant."""
    ss = (f * f).sum(axis=1, keepdims=True)
    dot = (y * dy).sum(axis=1, keepdims=True)
    active = ss > f.dtype.type(1e-12)
    return np.where(active, (dy - y * dot) * inv, dy * inv)


def head_logits(sig, wc, bc):
    """transpose [1,0,2] -> Flatten [B,62*256] (index k*256+d) -> Dense (nets/mj_uwyhNets_ba.py:848-850)."""
    bsz = sig.shape[1]
    flat = np.ascontiguousarray(sig.transpose(1, 0, 2)).reshape(bsz, -1)
    return flat @ wc + bc, flat


def softmax_xent(logits, onehot):
    """keras 'categorical_crossentropy' on a softmax Dense: TF2.3 graph mode back-tracks to the logits
    (nets/mj_uwyhNets_ba.py:865).  Returns (mean loss, probs)."""
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    return -(onehot * logp).sum(axis=1).mean(), np.exp(logp)


def batch_dist(x):
    """nets/triplet_loss_all.py:70-77.  x [n,m,d] -> [n,m,m]; exact 0 where the squared distance <= 0."""
    x2 = (x * x).sum(axis=2)
    d = x2[:, :, None] + x2[:, None, :] - x.dtype.type(2.0) * np.matmul(x, x.transpose(0, 2, 1))
    d = np.maximum(d, x.dtype.type(0))
    err = d <= 0
    d = np.sqrt(d + err.astype(x.dtype) * x.dtype.type(1e-16))
    return d * (~err).astype(x.dtype)


def triplet_index_lists(labels):
    """hp/hn index lists for one bin: flattened (i*m+j) pair indices in row-major order, exactly the
    boolean_mask order of nets/triplet_loss_all.py:40-47.  Pure integer function of the labels."""
    lab = np.asarray(labels).reshape(-1)
    m = lab.shape[0]
    same = (lab[:, None] == lab[None, :]).reshape(-1)
    hp = np.nonzero(same)[0].astype(np.int32)
    hn = np.nonzero(~same)[0].astype(np.int32)
    if hp.size % m or hn.size % m:
        raise ValueError("triplet_loss reshape([n,m,-1,1]) needs the positive/negative pair counts to be "
                         "divisible by the batch size (labels: %r)" % (lab.tolist(),))
    return hp, hn, hp.size // m, hn.size // m


def triplet_all(labels, emb, margin):
    """Batch-all triplet loss (nets/triplet_loss_all.py:8-67), literal flatten -> reshape semantics.
    labels [m], emb [n,m,d].  Returns (loss, aux) with aux carrying what the backward needs."""
    n, m, _ = emb.shape
    hp, hn, kp, kn = triplet_index_lists(labels)
    dist = batch_dist(emb).reshape(n, m * m)
    fhp = dist[:, hp].reshape(n, m, kp, 1)
    fhn = dist[:, hn].reshape(n, m, 1, kn)
    h = np.maximum(emb.dtype.type(margin) + (fhp - fhn), emb.dtype.type(0)).reshape(n, -1)
    ssum = h.sum(axis=1)
    num = (h > 0).astype(np.float32).sum(axis=1)
    with np.errstate(invalid='ignore', divide='ignore'):
        mean = np.where(num != 0, ssum / num.astype(emb.dtype), emb.dtype.type(0))
    return mean.mean(), dict(hp=hp, hn=hn, kp=kp, kn=kn, h=h, num=num, dist=dist.reshape(n, m, m))


def triplet_all_bwd(emb, aux, dloss=1.0):
    """SURVEY Appendix A.4-5.  N_k is a constant (tf.greater has no gradient)."""
    n, m, dd = emb.shape
    kp, kn = aux['kp'], aux['kn']
    act = (aux['h'] > 0).reshape(n, m, kp, kn).astype(emb.dtype)
    num = aux['num'].astype(emb.dtype)
    scale = np.where(num != 0, emb.dtype.type(dloss) / (np.maximum(num, 1) * n), 0).astype(emb.dtype)
    dmat = np.zeros((n, m * m), dtype=emb.dtype)
    ghp = act.sum(axis=3).reshape(n, m * kp) * scale[:, None]   # d/d full_hp_dist
    ghn = -act.sum(axis=2).reshape(n, m * kn) * scale[:, None]  # d/d full_hn_dist
    np.add.at(dmat, (slice(None), aux['hp']), ghp)
    np.add.at(dmat, (slice(None), aux['hn']), ghn)
    dmat = dmat.reshape(n, m, m)
    dist = aux['dist']
    with np.errstate(divide='ignore', invalid='ignore'):
        dq = np.where(dist > 0, dmat / (2 * dist), 0).astype(emb.dtype)   # d/d squared distance
    s = dq + dq.transpose(0, 2, 1)
    rows = s.sum(axis=2)
    # q_ij = r_i + r_j - 2 x_i.x_j  ->  dL/dx_i = 2*(rows_i x_i - sum_j s_ij x_j)
    return 2 * (rows[:, :, None] * emb - np.matmul(s, emb))


def triplet_hard(labels, emb, margin):
    """Batch-hard triplet loss per bin: tfa.losses.TripletHardLoss (soft=False, L2), the loss `compile_hard` names
    (nets/mj_uwyhNets_ba.py:1301-1306), restated from tensorflow_addons' published triplet_hard_loss and applied to each of the
    n bins of emb [n,m,d] (tfa itself takes [batch, dim]; the reference never calls compile_hard), mean over bins.
      pdist: as batch_dist, diagonal set to zero
      hard_positives = masked_maximum(pdist, adjacency - I)  = max((pdist - rowmin) * mask) + rowmin
      hard_negatives = masked_minimum(pdist, 1 - adjacency)  = min((pdist - rowmax) * mask) + rowmax
      loss = mean_a max(hp - hn + margin, 0)
    Returns (loss, aux)."""
    n, m, _ = emb.shape
    dt = emb.dtype.type
    lab = np.asarray(labels).reshape(-1)
    adj = lab[:, None] == lab[None, :]
    dist = batch_dist(emb) * (1 - np.eye(m, dtype=emb.dtype))[None]
    mpos = (adj & ~np.eye(m, dtype=bool)).astype(emb.dtype)[None]
    mneg = (~adj).astype(emb.dtype)[None]
    rmin = dist.min(axis=2, keepdims=True)
    rmax = dist.max(axis=2, keepdims=True)
    vp = (dist - rmin) * mpos
    vn = (dist - rmax) * mneg
    hp = vp.max(axis=2, keepdims=True) + rmin
    hn = vn.min(axis=2, keepdims=True) + rmax
    h = np.maximum(hp - hn + dt(margin), dt(0))[:, :, 0]
    return h.mean(axis=1).mean(), dict(dist=dist, vp=vp, vn=vn, mpos=mpos, mneg=mneg, h=h, num=(h > 0).sum(axis=1).astype(np.float32))


def triplet_hard_bwd(emb, aux, dloss=1.0):
    """Gradient of triplet_hard: reduce_max / reduce_min split their gradient equally among tied extrema (TF), the mask
    multiplies it, the row minimum / maximum receive the balance; then through the distances as in triplet_all_bwd."""
    n, m, _ = emb.shape
    dist, vp, vn, mpos, mneg = aux['dist'], aux['vp'], aux['vn'], aux['mpos'], aux['mneg']
    act = (aux['h'] > 0).astype(emb.dtype)[:, :, None] * emb.dtype.type(dloss / (m * n))
    ep = (vp == vp.max(axis=2, keepdims=True)).astype(emb.dtype)
    ep = ep / ep.sum(axis=2, keepdims=True)
    en = (vn == vn.min(axis=2, keepdims=True)).astype(emb.dtype)
    en = en / en.sum(axis=2, keepdims=True)
    emin = (dist == dist.min(axis=2, keepdims=True)).astype(emb.dtype)
    emin = emin / emin.sum(axis=2, keepdims=True)
    emax = (dist == dist.max(axis=2, keepdims=True)).astype(emb.dtype)
    emax = emax / emax.sum(axis=2, keepdims=True)
    gp = ep * mpos                     # d hp / d dist through (dist - rmin) * mask
    gp = gp + emin * (1 - gp.sum(axis=2, keepdims=True))      # ... and through rmin (-sum of the above + 1)
    gn = en * mneg
    gn = gn + emax * (1 - gn.sum(axis=2, keepdims=True))
    dmat = act * (gp - gn) * (1 - np.eye(m, dtype=emb.dtype))[None]
    with np.errstate(divide='ignore', invalid='ignore'):
        dq = np.where(dist > 0, dmat / (2 * dist), 0).astype(emb.dtype)
    s_ = dq + dq.transpose(0, 2, 1)
    rows = s_.sum(axis=2)
    return 2 * (rows[:, :, None] * emb - np.matmul(s_, emb))


def adam_step(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """keras Adam (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:227), t = 1-based step.
    As in TF's resource-apply kernel the hyper-parameters are cast to the variable dtype first, so
    (1 - beta) is formed in that dtype; lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) is a host scalar."""
    dt = p.dtype.type
    b1v, b2v, one = dt(b1), dt(b2), dt(1)
    m[...] = b1v * m + (one - b1v) * g
    v[...] = b2v * v + (one - b2v) * g * g
    lr_t = dt(lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t))
    p[...] = p - lr_t * m / (np.sqrt(v) + dt(eps))


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
CONV_SPECS = (  # name, k, cin (None = modality channels), cout
    ('a1', 5, None, 32), ('a2', 3, 32, 32),
    ('b1', 3, 32, 64), ('b2', 3, 64, 64),
    ('a3', 3, 32, 64), ('a4', 3, 64, 64),
    ('b3', 3, 64, 128), ('b4', 3, 128, 128),
    ('a5', 3, 64, 128), ('a6', 3, 128, 128),
)  # creation order of the Conv2D layers in nets/mj_uwyhNets_ba.py:428-462


def glorot_uniform(rng, shape, dtype=np.float32):
    """keras GlorotUniform; for rank>2 the receptive field is prod(shape[:-2])."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_branch_params(rng, cin, dtype=np.float32):
    p = {}
    for name, k, ci, co in CONV_SPECS:
        p[name] = glorot_uniform(rng, (k, k, cin if ci is None else ci, co), dtype)
    p['fc'] = glorot_uniform(rng, (NBINS, FEAT, HIDDEN), dtype)
    return p


def init_head_params(rng, nclasses, dtype=np.float32):
    return dict(wc=glorot_uniform(rng, (NBINS * HIDDEN, nclasses), dtype), bc=np.zeros((nclasses,), dtype))


# --------------------------------------------------------------------------------------
# encoder branch (nets/mj_uwyhNets_ba.py:419-484)
# --------------------------------------------------------------------------------------
def branch_forward(x, p):
    """x [B,L,60,60,C] -> (out [62,B,256], cache)."""
    bsz, L = x.shape[:2]
    c = {}
    xf = np.pad(x.reshape((bsz * L,) + x.shape[2:]), ((0, 0), (2, 2), (2, 2), (0, 0)))  # ZeroPadding2D(2) :428
    c['xf'] = xf
    c['a1'] = leaky(conv2d_same(xf, p['a1']))
    c['a2'] = leaky(conv2d_same(c['a1'], p['a2']))
    c['p2'], c['i2'] = maxpool2x2(c['a2'])                       # [N,32,32,32]
    p2_5 = c['p2'].reshape((bsz, L) + c['p2'].shape[1:])
    c['m1'] = setmax(p2_5)                                        # :435
    c['b1'] = leaky(conv2d_same(c['m1'], p['b1']))
    c['b2'] = leaky(conv2d_same(c['b1'], p['b2']))
    c['q2'], c['j2'] = maxpool2x2(c['b2'])                       # [B,16,16,64]
    c['a3'] = leaky(conv2d_same(c['p2'], p['a3']))
    c['a4'] = leaky(conv2d_same(c['a3'], p['a4']))
    c['p4'], c['i4'] = maxpool2x2(c['a4'])                       # [N,16,16,64]
    p4_5 = c['p4'].reshape((bsz, L) + c['p4'].shape[1:])
    c['m2'] = setmax(p4_5)                                        # :451
    c['s2'] = c['q2'] + c['m2']                                   # Add :452
    c['b3'] = leaky(conv2d_same(c['s2'], p['b3']))
    c['b4'] = leaky(conv2d_same(c['b3'], p['b4']))
    c['a5'] = leaky(conv2d_same(c['p4'], p['a5']))
    c['a6'] = leaky(conv2d_same(c['a5'], p['a6']))
    a6_5 = c['a6'].reshape((bsz, L) + c['a6'].shape[1:])
    c['m3'] = setmax(a6_5)                                        # :463  (branch_a)
    c['s3'] = c['b4'] + c['m3']                                   # :465  (branch_b)
    c['feat'] = hpp(c['m3'], c['s3'])                             # [62,B,128]
    out = binfc(c['feat'], p['fc'])
    c['B'], c['L'] = bsz, L
    return out, c


def branch_backward(dout, c, p):
    """dout [62,B,256] -> dict of parameter gradients (inputs need no gradient)."""
    bsz, L = c['B'], c['L']
    g = {}
    g['fc'], dfeat = binfc_bwd(c['feat'], p['fc'], dout)
    dm3, ds3 = hpp_bwd(c['m3'], c['s3'], dfeat)
    dm3 = dm3 + ds3                                               # m3 also feeds s3 = b4 + m3
    # global branch, block 2
    dz = leaky_bwd_from_out(c['b4'], ds3)
    g['b4'], d = conv2d_same_bwd(c['b3'], p['b4'], dz)
    dz = leaky_bwd_from_out(c['b3'], d)
    g['b3'], ds2 = conv2d_same_bwd(c['s2'], p['b3'], dz)
    dm2 = ds2
    dz = leaky_bwd_from_out(c['b2'], maxpool2x2_bwd(c['j2'], ds2))
    g['b2'], d = conv2d_same_bwd(c['b1'], p['b2'], dz)
    dz = leaky_bwd_from_out(c['b1'], d)
    g['b1'], dm1 = conv2d_same_bwd(c['m1'], p['b1'], dz)
    # frame stack, block 3
    a6_5 = c['a6'].reshape((bsz, L) + c['a6'].shape[1:])
    da6 = setmax_bwd(a6_5, c['m3'], dm3).reshape(c['a6'].shape)
    dz = leaky_bwd_from_out(c['a6'], da6)
    g['a6'], d = conv2d_same_bwd(c['a5'], p['a6'], dz)
    dz = leaky_bwd_from_out(c['a5'], d)
    g['a5'], dp4 = conv2d_same_bwd(c['p4'], p['a5'], dz)
    p4_5 = c['p4'].reshape((bsz, L) + c['p4'].shape[1:])
    dp4 = dp4 + setmax_bwd(p4_5, c['m2'], dm2).reshape(c['p4'].shape)
    # block 2
    dz = leaky_bwd_from_out(c['a4'], maxpool2x2_bwd(c['i4'], dp4))
    g['a4'], d = conv2d_same_bwd(c['a3'], p['a4'], dz)
    dz = leaky_bwd_from_out(c['a3'], d)
    g['a3'], dp2 = conv2d_same_bwd(c['p2'], p['a3'], dz)
    p2_5 = c['p2'].reshape((bsz, L) + c['p2'].shape[1:])
    dp2 = dp2 + setmax_bwd(p2_5, c['m1'], dm1).reshape(c['p2'].shape)
    # block 1
    dz = leaky_bwd_from_out(c['a2'], maxpool2x2_bwd(c['i2'], dp2))
    g['a2'], d = conv2d_same_bwd(c['a1'], p['a2'], dz)
    dz = leaky_bwd_from_out(c['a1'], d)
    g['a1'], _ = conv2d_same_bwd(c['xf'], p['a1'], dz, need_dx=False)
    return g


# --------------------------------------------------------------------------------------
# whole model (nets/mj_uwyhNets_ba.py:668-935 for 1/2 modalities, :1031-1299 for 3)
# --------------------------------------------------------------------------------------
def model_forward(xs, uses, params, mode='sign_max', multimodal=True):
    """xs: list of [B,L,60,60,C_m]; uses: list of [B,1] (ignored when not multimodal).
    Returns dict with signature [62,B,256], logits/probs [B,ncls] (if a head is present) and caches."""
    r = dict(branch=[], multimodal=multimodal, mode=mode)
    outs = []
    for x, bp in zip(xs, params['branches']):
        o, c = branch_forward(x, bp)
        outs.append(o)
        r['branch'].append(c)
    r['outs'] = outs
    if multimodal:
        gs = [gate(o, u) for o, u in zip(outs, uses)]
        f, sel = fuse(gs, mode)
        sig, inv = l2norm_batch(f)
        r.update(fused=f, sel=sel, inv=inv, uses=uses)
    else:
        sig = outs[0]   # single-modality graph: no gate, no normalisation (:893-903)
    r['signature'] = sig
    if 'head' in params:
        r['logits'], r['flat'] = head_logits(sig, params['head']['wc'], params['head']['bc'])
        z = r['logits'] - r['logits'].max(axis=1, keepdims=True)
        e = np.exp(z)
        r['probs'] = e / e.sum(axis=1, keepdims=True)
    return r


def model_loss_and_grads(xs, uses, labels, onehot, params, margin=0.2, loss_weights=(1.0, 0.1),
                         mode='sign_max', multimodal=True, triplet_mode='all'):
    """Total loss L = w0*triplet + w1*xent (nets/mj_uwyhNets_ba.py:865,933) and all parameter grads."""
    r = model_forward(xs, uses, params, mode, multimodal)
    sig = r['signature']
    dt = sig.dtype.type
    if triplet_mode == 'hard':      # the loss compile_hard would install (nets/mj_uwyhNets_ba.py:1301-1306)
        tri, aux = triplet_hard(labels, sig, margin)
        dsig = triplet_hard_bwd(sig, aux, dloss=loss_weights[0])
    else:
        tri, aux = triplet_all(labels, sig, margin)
        dsig = triplet_all_bwd(sig, aux, dloss=loss_weights[0])
    r['triplet'], r['tri_aux'] = tri, aux
    total = dt(loss_weights[0]) * tri
    grads = dict(branches=[])
    if 'head' in params:
        xent, probs = softmax_xent(r['logits'], onehot)
        r['xent'] = xent
        total = total + dt(loss_weights[1]) * xent
        bsz = onehot.shape[0]
        dlog = (probs - onehot) * dt(loss_weights[1] / bsz)
        grads['head'] = dict(wc=r['flat'].T @ dlog, bc=dlog.sum(axis=0))
        dflat = dlog @ params['head']['wc'].T
        dsig = dsig + dflat.reshape(bsz, NBINS, HIDDEN).transpose(1, 0, 2)
        r['acc'] = float((probs.argmax(axis=1) == onehot.argmax(axis=1)).mean())
    r['loss'] = total
    if multimodal:
        df = l2norm_batch_bwd(r['fused'], sig, r['inv'], dsig)
        dgs = fuse_bwd(r['sel'], df, len(xs), mode)
        douts = [gate(d, u) for d, u in zip(dgs, uses)]
    else:
        douts = [dsig]
    r['douts'] = douts
    for d, c, bp in zip(douts, r['branch'], params['branches']):
        grads['branches'].append(branch_backward(d.astype(sig.dtype), c, bp))
    return r, grads


def cast_params(params, dtype):
    out = dict(branches=[{k: v.astype(dtype) for k, v in bp.items()} for bp in params['branches']])
    if 'head' in params:
        out['head'] = {k: v.astype(dtype) for k, v in params['head'].items()}
    return out
