"""Routing census (test infrastructure): every discrete decision of the encoder path -- the argmax of the three MaxPools, of the
three set poolings over the frames, of the HPP strip maxima and the modality `sign_max` selects -- taken by the HIP path,
compared with the decisions of the fp64 oracle (oracle/torch_ref.py's graph, re-stated here with taps on the pre-decision values).

Why: a parameter gradient depends on WHERE these decisions route it.  Where two candidates agree to within the rounding of fp32
arithmetic the fp32-class HIP path and the fp64 oracle may legitimately decide differently ("flip"), and one flip under a dense
cotangent moves a whole tensor's relative-L2 error to 1e-3 ... 1e-2.  The tests therefore (1) COUNT the flips per layer, (2) prove
each one is a near-tie -- the value the oracle computes at the HIP path's choice is within `tol` (in units of the tensor's scale) of
the oracle's maximum -- and (3) compare gradients with the oracle forced to the HIP path's routing (`forced_branch`).

Reference semantics: MaxPool 2x2 first maximum in row-major window order (TF MaxPoolGrad), reduce_max over the frame axis and over
the HPP strips (nets/mj_uwyhNets_ba.py:433,435,448,451,463,473-477), tf.argmax first index in sign_max
(mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:171-176).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import torch_ref as T

FP32_ULP = 2.0 ** -23
# windows / positions whose runner-up is within CAND of the maximum (tensor scale = 1) keep their full gap vector; a HIP decision
# that differs from the oracle's anywhere else is further than CAND from a tie and is reported with gap = inf
CAND = 1e-3


def _first_argmax(v, dim):
    """index of the FIRST maximum along dim (torch.argmax does not promise the first one on ties)."""
    n = v.shape[dim]
    mx = v.amax(dim=dim, keepdim=True)
    w = torch.arange(n, 0, -1, dtype=torch.float64).reshape([n if d == (dim % v.dim()) else 1 for d in range(v.dim())])
    return ((v == mx).double() * w).argmax(dim=dim)


class Decision:
    """One family of decisions (a layer): the oracle's choice per decision, the tensor scale, and -- for the decisions whose
    runner-up is within CAND of the maximum -- the gap (max - value) of every candidate."""

    def __init__(self, name, values, dim):
        # values: fp64 tensor, candidates along `dim`; everything else = one decision per element
        v = values.movedim(dim, -1).contiguous()
        self.name = name
        self.scale = float(v.abs().max())
        self.ncand = v.shape[-1]
        self.shape = tuple(v.shape[:-1])
        mx = v.amax(dim=-1, keepdim=True)
        self.choice = _first_argmax(v, -1).to(torch.uint8).numpy().reshape(-1)
        gaps = (mx - v).reshape(-1, self.ncand)
        second = torch.topk(gaps, 2, dim=1, largest=False).values[:, 1]
        near = (second <= CAND * max(self.scale, 1e-300)).nonzero().reshape(-1)
        self.near_index = near.numpy()
        self.near_gaps = (gaps[near] / max(self.scale, 1e-300)).numpy()          # [k, ncand] in units of the scale
        self.nties = int((second == 0).sum())                                   # exact ties in the oracle itself

    def compare(self, hip_choice, active=None):
        """hip_choice: uint8 array of `shape` (the candidate the HIP path routed to; for reduce_max families ANY candidate that
        attains the HIP path's maximum is passed as a boolean mask [..., ncand] instead).  active: bool per clip (the leading axis
        of `shape` is clips or clips * frames, clip-major) -- decisions of inactive clips (a masked modality: the gate multiplies
        the branch by 0, its routing reaches no gradient) are not counted.  Returns (decisions, flips, worst gap)."""
        hip = np.asarray(hip_choice)
        n = int(np.prod(self.shape))
        keep = None
        if active is not None:
            active = np.asarray(active, bool)
            keep = np.repeat(active, n // active.size)
        if hip.dtype == np.bool_:                       # mask of the candidates that hold the HIP path's maximum
            mask = hip.reshape(n, self.ncand)
            agree = mask[np.arange(mask.shape[0]), self.choice]     # the oracle's choice is among them: no flip
            flips = np.nonzero(~agree)[0]
            pick = mask.argmax(axis=1)                  # (a flipped decision: the first candidate the HIP path routes to)
        else:
            pick = hip.reshape(n).astype(np.int64)
            flips = np.nonzero(pick != self.choice)[0]
        if keep is not None:
            flips = flips[keep[flips]]
            n = int(keep.sum())
        if flips.size == 0:
            return n, 0, 0.0
        pos = np.searchsorted(self.near_index, flips)
        pos_ok = (pos < self.near_index.size)
        found = np.zeros(flips.size, bool)
        found[pos_ok] = self.near_index[pos[pos_ok]] == flips[pos_ok]
        gap = np.full(flips.size, np.inf)
        gap[found] = self.near_gaps[pos[found], pick[flips[found]]]
        return n, int(flips.size), float(gap.max())


class SignDecision:
    """LeakyReLU's decision per element (slope 1 for x > 0, else 0.3: the TF rule `features > 0 ? g : alpha * g`): the oracle's sign,
    and for elements within CAND of zero (tensor scale = 1) their magnitude.  Stored NHWC like the HIP path's tensors."""

    def __init__(self, name, values_nchw):
        v = values_nchw.permute(0, 2, 3, 1).contiguous()
        self.name = name
        self.scale = float(v.abs().max())
        self.shape = tuple(v.shape)
        self.pos = (v > 0).numpy().reshape(-1)
        near = (v.abs().reshape(-1) <= CAND * max(self.scale, 1e-300)).nonzero().reshape(-1)
        self.near_index = near.numpy()
        self.near_mag = (v.reshape(-1)[near].abs() / max(self.scale, 1e-300)).numpy()

    def compare(self, hip_pos, active=None):
        hip = np.asarray(hip_pos, bool).reshape(-1)
        n = hip.size
        flips = np.nonzero(hip != self.pos)[0]
        if active is not None:
            keep = np.repeat(np.asarray(active, bool), n // np.asarray(active).size)
            flips = flips[keep[flips]]
            n = int(keep.sum())
        if flips.size == 0:
            return n, 0, 0.0
        pos = np.searchsorted(self.near_index, flips)
        ok = pos < self.near_index.size
        found = np.zeros(flips.size, bool)
        found[ok] = self.near_index[pos[ok]] == flips[ok]
        gap = np.full(flips.size, np.inf)
        gap[found] = self.near_mag[pos[found]]
        return n, int(flips.size), float(gap.max())


SIGN_KEYS = ('a1', 'p2', 'b1', 'q2', 'a3', 'p4', 'b3', 'b4', 'a5', 'a6')     # the ten LeakyReLU outputs of a branch, HIP buffer names


def branch_tapped(x, p, dec=None):
    """oracle/torch_ref.py `branch`, statement for statement, with autograd intact; `dec` (a dict) receives a Decision per routing
    family and a SignDecision per LeakyReLU, built from detached intermediates.  tests/test_routing_census.py checks that the
    outputs equal T.branch's to 1e-12.  Layouts of the decisions (what `hip_routing` must match):
      i2 / i4 / j2 : [N, C, H/2, W/2] windows, candidates = position dy * 2 + dx
      m1 / m2 / m3 : [B, C, H, W] set maxima, candidates = the L frames
      hpp_a / hpp_b: list over the 5 bin counts, [B, C, nb] strips, candidates = the 256 / nb positions of the strip
      sg_<name>    : NHWC, the sign of the LeakyReLU OUTPUT the HIP path saves (pooled layers: of the pooled value)"""
    bsz, L = x.shape[:2]
    tap = dec is not None

    def lrelu(t, name, pooled=False):
        y = F.leaky_relu(t, T.ALPHA)
        if tap and not pooled:
            dec['sg_' + name] = SignDecision('sg_' + name, y.detach())
        return y

    def pool(a, name, sname):
        n, c, h, w = a.shape
        if tap:
            win = a.detach().reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
            dec[name] = Decision(name, win, 4)
        y = F.max_pool2d(a, 2)
        if tap:
            dec['sg_' + sname] = SignDecision('sg_' + sname, y.detach())
        return y

    def setmax(a, name):
        v = a.reshape((bsz, L) + tuple(a.shape[1:]))
        if tap:
            dec[name] = Decision(name, v.detach(), 1)
        return torch.amax(v, dim=1)

    xf = F.pad(x.reshape((bsz * L,) + tuple(x.shape[2:])).permute(0, 3, 1, 2), (2, 2, 2, 2))
    a = lrelu(T._conv(xf, p['a1']), 'a1')
    a = pool(lrelu(T._conv(a, p['a2']), 'a2', pooled=True), 'i2', 'p2')
    b = setmax(a, 'm1')
    b = lrelu(T._conv(b, p['b1']), 'b1')
    b = pool(lrelu(T._conv(b, p['b2']), 'b2', pooled=True), 'j2', 'q2')
    a = lrelu(T._conv(a, p['a3']), 'a3')
    a = pool(lrelu(T._conv(a, p['a4']), 'a4', pooled=True), 'i4', 'p4')
    b = b + setmax(a, 'm2')
    b = lrelu(T._conv(b, p['b3']), 'b3')
    b = lrelu(T._conv(b, p['b4']), 'b4')
    a = lrelu(T._conv(a, p['a5']), 'a5')
    a = lrelu(T._conv(a, p['a6']), 'a6')
    a = setmax(a, 'm3')
    b = b + a
    feats = []
    for nb in T.BINS:
        for t, nm in ((a, 'hpp_a'), (b, 'hpp_b')):
            r = t.reshape(bsz, t.shape[1], nb, -1)
            if tap:
                dec.setdefault(nm, []).append(Decision('%s/%d' % (nm, nb), r.detach(), 3))
            feats.append((r.mean(dim=3) + torch.amax(r, dim=3)).permute(0, 2, 1))
    return torch.matmul(torch.cat(feats, dim=1).permute(1, 0, 2), p['fc'])


def oracle_branch_census(x, p):
    """branch_tapped without autograd: (out [62,B,256], {name: Decision | SignDecision})."""
    dec = {}
    with torch.no_grad():
        out = branch_tapped(x, p, dec)
    return out, dec


def sign_max_census(outs, uses):
    """The modality select of sign_max on the oracle's gated branch outputs: candidates = modalities, value = |g_m|."""
    with torch.no_grad():
        gs = torch.stack([o * u.reshape(1, -1, 1) for o, u in zip(outs, uses)], 0).abs()       # [M, 62, B, 256]
    return Decision('sel', gs, 0)


def _h2_or(t):
    return t.numpy() if hasattr(t, "meta") else t.cpu().numpy()


def hip_routing(core, mi):
    """The HIP path's saved tensors of branch `mi` after a forward pass: argmax bytes of the three MaxPools (NHWC), the stored frame
    values that the set poolings compared, the two HPP inputs.  Works for the f32 and the h2 path."""
    enc = core.encoders[mi]
    bufs = enc.h2.bufs if core.h2 else enc.act
    route = {k: bufs[k].cpu().numpy() for k in ('i2', 'i4', 'j2')}
    route.update({k: _h2_or(bufs[k]).astype(np.float64) for k in ('p2', 'p4', 'a6')})
    route.update({k: bufs[k].cpu().numpy().astype(np.float64) for k in ('m3', 's3')})
    # LeakyReLU decisions: the sign of every saved LeakyReLU output (what the backward pass reads its slope from); the first layer's
    # from its sign words (bit c of a pixel's word = a1[..., c] > 0)
    words = bufs['a1s'].cpu().numpy().astype(np.uint32)
    route['sg_a1'] = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    for k in SIGN_KEYS[1:]:
        route['sg_' + k] = route[k] > 0 if k in route else _h2_or(bufs[k]) > 0
    return route


def route_from_torch(x, p):
    """The same saved tensors as `hip_routing`, produced by the torch graph in the dtype of x (CPU stand-in for the HIP path: the
    helper's own test runs it in fp32 against the fp64 census)."""
    route = {}
    with torch.no_grad():
        bsz, L = x.shape[:2]
        lrelu = lambda t: F.leaky_relu(t, T.ALPHA)
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().numpy()

        def pool(a, name):
            n, c, h, w = a.shape
            win = a.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
            route[name] = nhwc(_first_argmax(win.double(), 4).to(torch.uint8))
            return win.amax(dim=4)

        def setmax(a, name):
            route[name] = nhwc(a).astype(np.float64)
            return a.reshape((bsz, L) + tuple(a.shape[1:])).amax(dim=1)

        xf = F.pad(x.reshape((bsz * L,) + tuple(x.shape[2:])).permute(0, 3, 1, 2), (2, 2, 2, 2))
        a = lrelu(T._conv(xf, p['a1']))
        a = pool(lrelu(T._conv(a, p['a2'])), 'i2')
        b = setmax(a, 'p2')
        b = lrelu(T._conv(b, p['b1']))
        b = pool(lrelu(T._conv(b, p['b2'])), 'j2')
        a = lrelu(T._conv(a, p['a3']))
        a = pool(lrelu(T._conv(a, p['a4'])), 'i4')
        b = b + setmax(a, 'p4')
        b = lrelu(T._conv(b, p['b3']))
        b = lrelu(T._conv(b, p['b4']))
        a = lrelu(T._conv(a, p['a5']))
        a = lrelu(T._conv(a, p['a6']))
        a = setmax(a, 'a6')
        b = b + a
        route['m3'], route['s3'] = nhwc(a).astype(np.float64), nhwc(b).astype(np.float64)
        dec = {}
        branch_tapped(x, p, dec)
        for k in SIGN_KEYS:
            route['sg_' + k] = dec['sg_' + k].pos.reshape(dec['sg_' + k].shape)
    return route


def census(dec, route, bsz, L, active=None):
    """{family: (decisions, flips, worst gap / scale)} of one branch: oracle decisions `dec` against the HIP path's `route`;
    active: bool per clip (None: all) -- see Decision.compare."""
    res = {}
    for k in ('i2', 'j2', 'i4'):
        res[k] = dec[k].compare(np.transpose(route[k], (0, 3, 1, 2)), active)         # NHWC -> [N, C, H/2, W/2]
    for k, src in (('m1', 'p2'), ('m2', 'p4'), ('m3', 'a6')):
        v = route[src].reshape((bsz, L) + route[src].shape[1:])                       # [B, L, H, W, C]: the values the HIP path compared
        mask = v == v.max(axis=1, keepdims=True)
        res[k] = dec[k].compare(np.transpose(mask, (0, 4, 2, 3, 1)), active)          # -> [B, C, H, W, L]
    for nm, src in (('hpp_a', 'm3'), ('hpp_b', 's3')):
        tot, flips, worst = 0, 0, 0.0
        t = np.transpose(route[src], (0, 3, 1, 2))                                    # [B, C, 16, 16]
        for d, nb in zip(dec[nm], T.BINS):
            r = t.reshape(t.shape[0], t.shape[1], nb, -1)
            n, f, w = d.compare(r == r.max(axis=3, keepdims=True), active)
            tot, flips, worst = tot + n, flips + f, max(worst, w)
        res[nm] = (tot, flips, worst)
    tot, flips, worst = 0, 0, 0.0
    for k in SIGN_KEYS:
        n, f, w = dec['sg_' + k].compare(route['sg_' + k], active)
        tot, flips, worst = tot + n, flips + f, max(worst, w)
    res['lrelu'] = (tot, flips, worst)
    return res


def forced_branch(x, p, route):
    """oracle/torch_ref.py `branch` with every routing decision -- MaxPool argmax, set-max over the frames, the strip maximum of
    HPP -- taken from the HIP path (`route`, see hip_routing) instead of from the oracle's own values: what is left to differ from
    the HIP path's gradients is arithmetic.  Differentiable (fp64 autograd)."""
    bsz, L = x.shape[:2]

    def lrelu(t, key):                      # slope from the HIP path's saved sign (NHWC bool), not from the oracle's own value
        pos = torch.from_numpy(np.ascontiguousarray(route['sg_' + key])).permute(0, 3, 1, 2)
        return torch.where(pos, t, T.ALPHA * t)

    def pool(a, idx):                       # a [N,C,H,W]; idx [N,H/2,W/2,C] position dy * 2 + dx inside the 2 x 2 window
        n, c, h, w = a.shape
        win = a.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
        ix = torch.from_numpy(idx.astype(np.int64)).permute(0, 3, 1, 2).unsqueeze(-1)
        return torch.gather(win, 4, ix).squeeze(-1)

    def setmax(a, vals):                    # a [B*L,C,H,W]; vals: the HIP path's stored frame values [B*L,H,W,C]
        v = torch.from_numpy(np.asarray(vals, dtype=np.float64)).reshape((bsz, L) + tuple(vals.shape[1:]))
        m = (v == v.amax(dim=1, keepdim=True)).double()
        m = (m / m.sum(dim=1, keepdim=True)).permute(0, 1, 4, 2, 3)
        return (a.reshape((bsz, L) + tuple(a.shape[1:])) * m).sum(dim=1)

    xf = F.pad(x.reshape((bsz * L,) + tuple(x.shape[2:])).permute(0, 3, 1, 2), (2, 2, 2, 2))
    # (a pooled layer: the routed pre-activation sum first, LeakyReLU with the sign of the saved pooled value on the winner only --
    #  LeakyReLU is increasing, so pooling before or after it selects the same element)
    a = lrelu(T._conv(xf, p['a1']), 'a1')
    a = lrelu(pool(T._conv(a, p['a2']), route['i2']), 'p2')
    b = setmax(a, route['p2'])
    b = lrelu(T._conv(b, p['b1']), 'b1')
    b = lrelu(pool(T._conv(b, p['b2']), route['j2']), 'q2')
    a = lrelu(T._conv(a, p['a3']), 'a3')
    a = lrelu(pool(T._conv(a, p['a4']), route['i4']), 'p4')
    b = b + setmax(a, route['p4'])
    b = lrelu(T._conv(b, p['b3']), 'b3')
    b = lrelu(T._conv(b, p['b4']), 'b4')
    a = lrelu(T._conv(a, p['a5']), 'a5')
    a = lrelu(T._conv(a, p['a6']), 'a6')
    a = setmax(a, route['a6'])
    b = b + a
    feats = []
    for nb in T.BINS:
        for t, hv in ((a, route['m3']), (b, route['s3'])):
            r = t.reshape(bsz, t.shape[1], nb, -1)
            rh = torch.from_numpy(np.asarray(hv, dtype=np.float64)).permute(0, 3, 1, 2).reshape(bsz, t.shape[1], nb, -1)
            m = (rh == rh.amax(dim=3, keepdim=True)).double()
            m = m / m.sum(dim=3, keepdim=True)
            feats.append((r.mean(dim=3) + (r * m).sum(dim=3)).permute(0, 2, 1))
    return torch.matmul(torch.cat(feats, dim=1).permute(1, 0, 2), p['fc'])


def forced_step_grads(x64, u64, labels, onehot, p64, routes, sel, margin=0.2, loss_weights=(1.0, 0.1), multimodal=True, mode='sign_max'):
    """The whole step's parameter gradients with the fp64 oracle forced to the HIP path's routing: MaxPool / set-max / HPP decisions
    from `routes` (hip_routing per branch), the modality select of sign_max from `sel` ([62,B,256] uint8; None: single modality).
    x64 / u64: fp64 tensors; p64: numpy parameter dict (oracle layout).  Returns numpy gradients in the oracle's layout."""
    tp = T.params_from_numpy(p64, dtype=torch.float64)
    leaves = [v for bp in tp["branches"] for v in bp.values()] + list(tp["head"].values())
    for t in leaves:
        t.requires_grad_(True)
    outs = [forced_branch(x, bp, r) for x, bp, r in zip(x64, tp["branches"], routes)]
    if multimodal:
        gs = torch.stack([o * u.reshape(1, -1, 1) for o, u in zip(outs, u64)], 0)
        if mode == 'avg':
            f = gs.mean(dim=0)
        else:           # sign_max / max: the modality the HIP path selected per element
            f = torch.gather(gs, 0, torch.from_numpy(np.asarray(sel).astype(np.int64)).unsqueeze(0)).squeeze(0)
        sig = f * torch.rsqrt((f * f).sum(dim=1, keepdim=True).clamp_min(1e-12))
    else:
        sig = outs[0]
    oh = torch.from_numpy(np.asarray(onehot, dtype=np.float64))
    logits = sig.permute(1, 0, 2).reshape(sig.shape[1], -1) @ tp["head"]["wc"] + tp["head"]["bc"]
    total = loss_weights[0] * T.triplet(torch.from_numpy(np.asarray(labels)), sig, margin) + \
        loss_weights[1] * (-(oh * F.log_softmax(logits, dim=1)).sum(dim=1).mean())
    total.backward()
    return dict(branches=[{k: v.grad.numpy() for k, v in bp.items()} for bp in tp["branches"]],
                head={k: v.grad.numpy() for k, v in tp["head"].items()})


def grad_errors(got, ref):
    """relative L2 per parameter tensor, {'m<i>.<name>' | 'head.<name>': error}"""
    rl2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))
    worst = {}
    for mi, bp in enumerate(ref["branches"]):
        for k, v in bp.items():
            worst["m%d.%s" % (mi, k)] = rl2(got["branches"][mi][k], np.asarray(v, np.float64))
    for k, v in ref.get("head", {}).items():
        worst["head." + k] = rl2(got["head"][k], np.asarray(v, np.float64))
    return worst


def format_census(res):
    return "; ".join("%s %d/%d flips%s" % (k, f, n, "" if f == 0 else " (worst gap %.2g ulp32 of scale)" % (w / FP32_ULP))
                     for k, (n, f, w) in res.items())


def check_gradients(core, ref_grads, xs, uses, labels, onehot, p64, tight, mode='sign_max', margin=0.2, loss_weights=(1.0, 0.1),
                    multimodal=True, near_tie=8, forced_bar=5e-5, label="", loose=None):
    """The gradient bar of the end-to-end tests, stated through the census.  Every parameter tensor within
    `tight` (relative L2) of the oracle's gradient, OR -- when a tensor is beyond it -- (1) every MaxPool / set-max / HPP / LeakyReLU
    decision in which the HIP path differs from the fp64 oracle is a near-tie (`near_tie` fp32 ulp of the tensor's scale), and (2) the
    oracle forced to the HIP path's decisions gives the HIP gradients to `forced_bar`.  Returns (worst error, number of flips or None)."""
    got = core.get_grads_numpy()
    worst = grad_errors(got, ref_grads)
    if max(worst.values()) <= tight:
        return max(worst.values()), None
    if loose is not None:         # (a sibling case of the same encoders carries the census: only the flat bar here)
        assert max(worst.values()) <= loose, (label, worst)
        return max(worst.values()), None
    nmod = len(core.encoders)
    bsz, L = xs[0].shape[0], xs[0].shape[1]
    routes = [hip_routing(core, mi) for mi in range(nmod)]
    flips, lines = 0, []
    for mi in range(nmod):
        tp = {k: torch.from_numpy(np.asarray(v, dtype=np.float64)) for k, v in p64['branches'][mi].items()}
        _, dec = oracle_branch_census(torch.from_numpy(np.asarray(xs[mi], dtype=np.float64)), tp)
        active = (np.asarray(uses[mi]).reshape(-1) != 0) if multimodal else None
        res = census(dec, routes[mi], bsz, L, active)
        lines.append("m%d: %s" % (mi, format_census({k: v for k, v in res.items() if v[1]})))
        for fam, (n, f, w) in res.items():
            assert w <= near_tie * FP32_ULP, "%s branch %d, %s: %d of %d decisions differ, the worst %.3g fp32 ulp from a tie" % (
                label, mi, fam, f, n, w / FP32_ULP)
            flips += f
    x64 = [torch.from_numpy(np.asarray(x, dtype=np.float64)) for x in xs]
    u64 = [torch.from_numpy(np.asarray(u, dtype=np.float64)) for u in uses] if multimodal else None
    sel = core.sel.cpu().numpy() if (multimodal and mode != 'avg') else None
    gf = forced_step_grads(x64, u64, labels, onehot, p64, routes, sel, margin, loss_weights, multimodal, mode)
    wf = grad_errors(got, gf)
    print("%s worst gradient tensor %.2e (%s) against the oracle's own routing; %d decisions differ, all near-ties (%s); against the "
          "oracle forced to the HIP path's decisions: worst %.2e (%s)"
          % (label, max(worst.values()), max(worst, key=worst.get), flips, " | ".join(lines), max(wf.values()), max(wf, key=wf.get)))
    assert max(wf.values()) <= forced_bar, (label, wf)
    return max(worst.values()), flips
