"""Independent torch-CPU (autograd) statement of the same graph as oracle/ugaitnet_oracle.py.

TEST INFRASTRUCTURE ONLY (see the header of ugaitnet_oracle.py: parity is UNPINNED by the reference,
so the numpy restatement is pinned against this second, independently written implementation).
It is also the `cpu_baseline` leg of bench.py (kind "port": oneDNN-backed torch CPU ops, fwd+bwd+Adam).

Nothing here shares code with the numpy oracle: convolutions are F.conv2d (NCHW), pooling is
F.max_pool2d, every gradient comes from autograd.
Reference call sites: nets/mj_uwyhNets_ba.py:419-484 (encoder), :23-54 (MatMul, gate), :814-851 /
:1189-1214 (fusion, signature, head), nets/triplet_loss_all.py:8-77,
mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:169-178 (sign_max).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

ALPHA = 0.3
BINS = (1, 2, 4, 8, 16)


def _conv(x, w):
    # x NCHW, w HWIO -> OIHW
    k = w.shape[0]
    return F.conv2d(x, w.permute(3, 2, 0, 1), padding=k // 2)


def _lrelu(x):
    return F.leaky_relu(x, ALPHA)


def branch(x, p):
    """x [B,L,60,60,C] -> [62,B,256]."""
    bsz, L = x.shape[:2]
    xf = x.reshape((bsz * L,) + tuple(x.shape[2:])).permute(0, 3, 1, 2)
    xf = F.pad(xf, (2, 2, 2, 2))
    a = _lrelu(_conv(xf, p['a1']))
    a = _lrelu(_conv(a, p['a2']))
    a = F.max_pool2d(a, 2)
    b = torch.amax(a.reshape((bsz, L) + tuple(a.shape[1:])), dim=1)
    b = _lrelu(_conv(b, p['b1']))
    b = _lrelu(_conv(b, p['b2']))
    b = F.max_pool2d(b, 2)
    a = _lrelu(_conv(a, p['a3']))
    a = _lrelu(_conv(a, p['a4']))
    a = F.max_pool2d(a, 2)
    b = b + torch.amax(a.reshape((bsz, L) + tuple(a.shape[1:])), dim=1)
    b = _lrelu(_conv(b, p['b3']))
    b = _lrelu(_conv(b, p['b4']))
    a = _lrelu(_conv(a, p['a5']))
    a = _lrelu(_conv(a, p['a6']))
    a = torch.amax(a.reshape((bsz, L) + tuple(a.shape[1:])), dim=1)  # [B,128,16,16]
    b = b + a
    feats = []
    for nb in BINS:
        for t in (a, b):
            r = t.reshape(bsz, t.shape[1], nb, -1)       # NCHW: H*W row-major -> strips
            feats.append((r.mean(dim=3) + torch.amax(r, dim=3)).permute(0, 2, 1))  # [B,nb,C]
    feat = torch.cat(feats, dim=1).permute(1, 0, 2)        # [62,B,128]
    return torch.matmul(feat, p['fc'])


def sign_max(gs):
    st = torch.stack(gs, 0)
    flat = st.reshape(len(gs), -1)
    pos = torch.argmax(flat.abs(), dim=0, keepdim=True)
    return torch.gather(flat, 0, pos).reshape(gs[0].shape)


def keras_maximum(gs):
    out = gs[0]
    for g in gs[1:]:
        out = torch.where(out >= g, out, g)   # tf.maximum gradient rule: ties go to the first argument
    return out


def batch_dist(x):
    x2 = (x * x).sum(dim=2)
    d = x2[:, :, None] + x2[:, None, :] - 2.0 * torch.matmul(x, x.transpose(1, 2))
    d = torch.clamp_min(d, 0.0)
    err = d <= 0.0
    d = torch.sqrt(d + err.to(x.dtype) * 1e-16)
    return d * (~err).to(x.dtype)


def triplet(labels, emb, margin, return_counts=False):
    """return_counts: also the per-bin number of active (> 0) hinges, the `num` of nets/triplet_loss_all.py:54-55."""
    n, m, _ = emb.shape
    lab = labels.reshape(1, m).repeat(n, 1)
    hp = (lab[:, None, :] == lab[:, :, None]).reshape(-1)
    hn = ~hp
    d = batch_dist(emb).reshape(-1)
    fhp = d[hp].reshape(n, m, -1, 1)
    fhn = d[hn].reshape(n, m, 1, -1)
    h = torch.clamp_min(margin + (fhp - fhn), 0.0).reshape(n, -1)
    s = h.sum(dim=1)
    num = (h > 0).to(torch.float32).sum(dim=1)
    mean = torch.where(num != 0, s / num.to(s.dtype).clamp_min(1.0), torch.zeros_like(s))
    if return_counts:
        return mean.mean(), num.detach()
    return mean.mean()


def triplet_hard(labels, emb, margin):
    """tfa.losses.TripletHardLoss (soft=False, L2) per bin, written with torch ops the way tensorflow_addons writes it
    (pairwise_distance with a zeroed diagonal, _masked_maximum / _masked_minimum); autograd supplies the gradient."""
    n, m, _ = emb.shape
    lab = labels.reshape(m, 1)
    adj = lab == lab.t()
    eye = torch.eye(m, dtype=emb.dtype)
    pd = batch_dist(emb) * (1.0 - eye)
    mneg = (~adj).to(emb.dtype)
    mpos = adj.to(emb.dtype) - eye
    rmax = pd.amax(dim=2, keepdim=True)
    hn = ((pd - rmax) * mneg).amin(dim=2, keepdim=True) + rmax
    rmin = pd.amin(dim=2, keepdim=True)
    hp = ((pd - rmin) * mpos).amax(dim=2, keepdim=True) + rmin
    return torch.clamp_min(hp - hn + margin, 0.0).mean(dim=(1, 2)).mean()


def forward(xs, uses, params, mode='sign_max', multimodal=True, branch_fn=None):
    # (branch_fn: a statement-for-statement copy of `branch` that also records its routing decisions, tests/routing.py)
    outs = [(branch_fn or branch)(x, bp) for x, bp in zip(xs, params['branches'])]
    if multimodal:
        gs = [o * u.reshape(1, -1, 1) for o, u in zip(outs, uses)]
        if mode == 'sign_max':
            f = sign_max(gs)
        elif mode == 'max':
            f = keras_maximum(gs)
        else:
            f = torch.stack(gs, 0).mean(dim=0)
        ss = (f * f).sum(dim=1, keepdim=True).clamp_min(1e-12)
        sig = f * torch.rsqrt(ss)
    else:
        sig = outs[0]
    res = dict(signature=sig, outs=outs)
    if 'head' in params:
        flat = sig.permute(1, 0, 2).reshape(sig.shape[1], -1)
        res['logits'] = flat @ params['head']['wc'] + params['head']['bc']
    return res


def loss_and_grads(xs, uses, labels, onehot, params, margin=0.2, loss_weights=(1.0, 0.1),
                   mode='sign_max', multimodal=True, branch_fn=None):
    leaves = []
    for bp in params['branches']:
        leaves += list(bp.values())
    if 'head' in params:
        leaves += list(params['head'].values())
    for t in leaves:
        t.requires_grad_(True)
        t.grad = None
    res = forward(xs, uses, params, mode, multimodal, branch_fn=branch_fn)
    tri, res['tri_counts'] = triplet(labels, res['signature'], margin, return_counts=True)
    total = loss_weights[0] * tri
    res['triplet'] = tri.detach()
    if 'head' in params:
        xent = -(onehot * F.log_softmax(res['logits'], dim=1)).sum(dim=1).mean()
        res['xent'] = xent.detach()
        total = total + loss_weights[1] * xent
    total.backward()
    res['loss'] = total.detach()
    grads = dict(branches=[{k: v.grad for k, v in bp.items()} for bp in params['branches']])
    if 'head' in params:
        grads['head'] = {k: v.grad for k, v in params['head'].items()}
    return res, grads


def params_from_numpy(params, dtype=torch.float32):
    out = dict(branches=[{k: torch.tensor(v, dtype=dtype) for k, v in bp.items()} for bp in params['branches']])
    if 'head' in params:
        out['head'] = {k: torch.tensor(v, dtype=dtype) for k, v in params['head'].items()}
    return out


class TorchTrainer:
    """fwd + bwd + Adam on the CPU, used as bench.py's cpu_baseline ("port")."""

    def __init__(self, params, lr=1e-4, margin=0.2, loss_weights=(1.0, 0.1), mode='sign_max', multimodal=True):
        self.params = params
        leaves = []
        for bp in params['branches']:
            leaves += list(bp.values())
        if 'head' in params:
            leaves += list(params['head'].values())
        for t in leaves:
            t.requires_grad_(True)
        self.opt = torch.optim.Adam(leaves, lr=lr, betas=(0.9, 0.999), eps=1e-7)
        self.kw = dict(margin=margin, loss_weights=loss_weights, mode=mode, multimodal=multimodal)

    def step(self, xs, uses, labels, onehot):
        self.opt.zero_grad(set_to_none=True)
        res = forward(xs, uses, self.params, self.kw['mode'], self.kw['multimodal'])
        total = self.kw['loss_weights'][0] * triplet(labels, res['signature'], self.kw['margin'])
        if 'head' in self.params:
            total = total + self.kw['loss_weights'][1] * (
                -(onehot * F.log_softmax(res['logits'], dim=1)).sum(dim=1).mean())
        total.backward()
        self.opt.step()
        return float(total.detach())
