"""bf16 tensors in HBM (BASELINE.json configs[4], SURVEY 8(d) "C5") and the host wrappers of the kernels that consume them.

A tensor [n, h, w, c] is held as bf16 bit patterns in an int16 torch tensor (half the bytes of fp32); the 3x3 layers multiply
them on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (csrc/conv3x3_bf.hip, wgrad3x3_bf.hip), the steps between them read and
write bf16 (csrc/bf_elem.hip).  Weight gradients, master weights and Adam stay fp32.  torch holds buffers only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, ptr_array

I16 = torch.int16
U8 = torch.uint8
F32 = torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _sizes(v):
    return (C.c_size_t * len(v))(*[int(x) for x in v])


def _opt(ts):
    return None if ts is None else ptr_array([None if t is None else t for t in ts])


def _hbm(kernel, nbytes):
    return dict(flops=None, mfma_flops=None, bytes=float(nbytes), kernel=kernel, bound="hbm")


def empty(shape, device):
    return torch.empty(tuple(shape), dtype=I16, device=device)


def from_f32(x, out=None):
    """fp32 tensor -> bf16 bit patterns (round to nearest even), one kernel."""
    out = empty(x.shape, x.device) if out is None else out
    convert_multi([x], [out])
    return out


def to_numpy(t):
    """bf16 bit patterns -> float64 numpy (tests)."""
    u = t.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    return u.view(np.float32).astype(np.float64)


def convert_multi(xs, outs):
    call("ugn_bf_convert_multi", ptr_array(xs), ptr_array(outs), _sizes([x.numel() for x in xs]), len(xs), _stream(),
         label="bf_convert", work=_hbm("cvt_multi_kernel", sum(x.numel() for x in xs) * 6.0))
    return outs


def pack_multi(jobs):
    """jobs: list of (w HWIO fp32, packed int16 tensor of 9*cin*cout elements, dgrad flag, layer-is-pooled flag)."""
    for k in range(0, len(jobs), 64):
        part = jobs[k:k + 64]
        n = len(part)
        call("ugn_bf_pack_multi", ptr_array([j[0] for j in part]), ptr_array([j[1] for j in part]),
             _ints([j[0].shape[2] for j in part]), _ints([j[0].shape[3] for j in part]), _ints([bool(j[2]) for j in part]),
             _ints([bool(j[3]) for j in part]), n, _stream())


def pack(w, dgrad, pooled=False):
    pk = torch.empty((9 * w.shape[2] * w.shape[3],), dtype=I16, device=w.device)
    pack_multi([(w, pk, dgrad, pooled)])
    return pk


def _work(kind, hw, cin, cout, pooled, ns, act=False):
    n = int(sum(ns))
    flops = 2.0 * 9 * cin * cout * hw * hw * n
    if kind == "fwd":
        kern = "conv_bf_kernel<%d, %d, %d, 0, %d>" % (cin, cout, hw, 1 if pooled else 0)
    elif kind == "dgrad":
        kern = "conv_bf_kernel<%d, %d, %d, %d, %d>" % (cout, cin, hw, int(pooled), 3 if act else 2)
    else:
        kern = "wgrad_bf_kernel<%d, %d, %d, %d>" % (cin, cout, hw, int(pooled))
    label = "conv3x3_%s[%d->%d @%dx%d%s bf16] %s" % (kind, cin, cout, hw, hw, " pooled" if pooled else "", kern)
    # Algorithmic HBM bytes (every tensor once): bf16 tensors at 2 B per element, argmax maps 1 B, fp32 weight
    # gradients negligible.  With bf16 operands the 3x3 layers sit far below the matrix roof (10-40 % busy): the HBM rate is the
    # figure bench.py reports first for them, the MFMA fraction second.
    full, pool_ = hw * hw, (hw // 2) * (hw // 2)
    if kind == "fwd":       # in: full-resolution input; out: (pooled) output (+ argmax bytes)
        nbytes = n * (full * cin * 2 + (pool_ * cout * 3 if pooled else full * cout * 2))
    elif kind == "dgrad":   # in: (pooled) output gradient (+ argmax bytes) (+ the activation for LeakyReLU'); out: input gradient
        nbytes = n * ((pool_ * cout * 3 if pooled else full * cout * 2) + full * cin * 2 + (full * cin * 2 if act else 0))
    else:                   # in: input + (pooled) output gradient (+ argmax bytes)
        nbytes = n * (full * cin * 2 + (pool_ * cout * 3 if pooled else full * cout * 2))
    return label, dict(flops=flops, mfma_flops=flops, bytes=float(nbytes), kernel=kern, bound="hbm", images=n, dtype="bf16")


def conv3x3_fwd_multi(xs, wpks, cout, pool, outs, idxs=None):
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    ho = hw // 2 if pool else hw
    for x, o in zip(xs, outs):
        assert x.dtype == I16 and tuple(o.shape) == (x.shape[0], ho, ho, cout), (x.shape, o.shape)
    label, work = _work("fwd", hw, cin, cout, pool, [x.shape[0] for x in xs])
    call("ugn_bf_conv3x3_fwd_multi", ptr_array(xs), ptr_array(wpks), ptr_array(outs), ptr_array(idxs) if pool else None,
         _ints([x.shape[0] for x in xs]), len(xs), hw, cin, cout, int(bool(pool)), _stream(), label=label, work=work)
    return (outs, idxs) if pool else outs


def conv3x3_dgrad_multi(dzs, wpks, hw, cin, cout, outs, dz_idxs=None, acts=None):
    for d, o in zip(dzs, outs):
        assert tuple(o.shape) == (d.shape[0], hw, hw, cin), (d.shape, o.shape)
    label, work = _work("dgrad", hw, cin, cout, dz_idxs is not None, [d.shape[0] for d in dzs], act=acts is not None)
    call("ugn_bf_conv3x3_dgrad_multi", ptr_array(dzs), _opt(dz_idxs), ptr_array(wpks), _opt(acts), ptr_array(outs),
         _ints([d.shape[0] for d in dzs]), len(dzs), hw, cin, cout, _stream(), label=label, work=work)
    return outs


_WS = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(max(int(nbytes), 1), dtype=U8, device=device)
    return buf


def conv3x3_wgrad_multi(xs, dzs, cout, dws, dz_idxs=None):
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    nbytes = _lib.load().ugn_bf_conv3x3_wgrad_ws(hw, cin, cout)
    if nbytes == 0:
        raise ValueError("bf16 conv3x3_wgrad_multi: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _workspace(nbytes, xs[0].device)
    label, work = _work("wgrad", hw, cin, cout, dz_idxs is not None, [x.shape[0] for x in xs])
    call("ugn_bf_conv3x3_wgrad_multi", ptr_array(xs), ptr_array(dzs), _opt(dz_idxs), ptr_array(dws), _ints([x.shape[0] for x in xs]),
         len(xs), hw, cin, cout, ptr(ws), ws.numel(), _stream(), label=label, work=work)
    return dws


def conv5x5_in_fwd(x, w, out, sign=None):
    n, cin = x.shape[0], x.shape[3]
    call("ugn_conv5x5_in_fwd_bf", ptr(x), ptr(w), ptr(out), ptr(sign), n, cin, _stream(), label="conv5x5_fwd[cin=%d bf16]" % cin,
         work=_hbm("conv5x5_fwd_kernel<%d, %s, 2>" % (cin, "true" if sign is not None else "false"),
                   n * (3600.0 * cin * 4 + 4096 * 32 * 2 + (4096 * 4 if sign is not None else 0))))
    return out


def conv5x5_in_wgrad(x, dz1, dw, sign=None):
    n, cin = x.shape[0], x.shape[3]
    nbytes = _lib.load().ugn_conv5x5_in_wgrad_ws(n, cin)
    ws = _workspace(nbytes, x.device)
    call("ugn_conv5x5_in_wgrad_bf", ptr(x), ptr(dz1), ptr(sign), ptr(dw), n, cin, ptr(ws), ws.numel(), _stream(),
         label="conv5x5_wgrad[cin=%d bf16]" % cin,
         work=_hbm("conv5x5_wgrad_kernel<%d, %s, 2>" % (cin, "true" if sign is not None else "false"),
                   n * (3600.0 * cin * 4 + 4096 * 32 * 2 + (4096 * 4 if sign is not None else 0))))
    return dw


def setmax_fwd_multi(ps, bs, l, ms=None, addends=None, sums=None, routes=None):
    """routes (optional): int32 tensors [b, h, w, 2, c] that receive the routing words (h2.setmax_fwd_h2_multi)."""
    n, h, w, c = ps[0].shape
    call("ugn_bf_setmax_fwd_routed_multi", ptr_array(ps), _opt(addends), _opt(ms), _opt(sums), _opt(routes), _ints(bs), len(ps), l, h * w, c,
         _stream(), label="setmax_fwd[%dx%dx%d bf16]" % (h, w, c),
         work=_hbm("setmax_fwd_bf_kernel<false>", sum(bs) * ((l + 1.0) * 2 + (8.0 if routes is not None else 0.0)) * h * w * c))
    return ms, sums


def setmax_fwd_f32_multi(ps, bs, l, ms, addends, sums, routes=None):
    n, h, w, c = ps[0].shape
    call("ugn_bf_setmax_fwd_f32_routed_multi", ptr_array(ps), _opt(addends), _opt(ms), _opt(sums), _opt(routes), _ints(bs), len(ps), l,
         h * w, c, _stream(), label="setmax_fwd[%dx%dx%d bf16->f32]" % (h, w, c),
         work=_hbm("setmax_fwd_bf_kernel<true>", sum(bs) * ((l + 2.0) * 2 + (8.0 if routes is not None else 0.0)) * h * w * c))
    return ms, sums


def setmax_bwd_multi(ps, dms, bs, l, lrelu, outs, addends=None, dm_is_f32=False, routes=None):
    """routes: the routing words of the forward pass -- with them the frames ps are not read (same results)."""
    n, h, w, c = outs[0].shape
    tail = (ptr_array(dms), int(bool(dm_is_f32)), _opt(addends), ptr_array(outs), _ints(bs), len(outs), l, h * w, c, int(bool(lrelu)),
            _stream())
    label = "setmax_bwd[%dx%dx%d bf16%s]" % (h, w, c, " +addend" if addends is not None else "")
    streams = 2 if addends is not None else 1
    if routes is not None:
        call("ugn_bf_setmax_bwd_routed_multi", ptr_array(routes), *tail, label=label,
             work=_hbm("setmax_bwd_bf_kernel<%s, true>" % ("true" if dm_is_f32 else "false"),
                       sum(bs) * h * w * c * (l * 2.0 * streams + 10.0)))
    else:
        call("ugn_bf_setmax_bwd_multi", ptr_array(ps), *tail, label=label,
             work=_hbm("setmax_bwd_bf_kernel<%s, false>" % ("true" if dm_is_f32 else "false"),
                       sum(bs) * l * h * w * c * 2.0 * (streams + 1)))
    return outs


def lrelu_bwd_multi(gs, acts, outs):
    n, h, w, c = gs[0].shape
    call("ugn_bf_lrelu_bwd_multi", ptr_array(gs), ptr_array(acts), ptr_array(outs), _sizes([g.shape[0] * h * w for g in gs]), len(gs), c,
         _stream(), label="lrelu_bwd[bf16]", work=_hbm("lrelu_bwd_bf_kernel", sum(g.numel() for g in gs) * 2.0 * 3))
    return outs


def hpp_bwd_b4_multi(as_, s3s, b4s, dfeats, dm3s, dzb4s):
    bs = [a.shape[0] for a in as_]
    call("ugn_hpp_bwd_b4bf_multi", ptr_array(as_), ptr_array(s3s), ptr_array(b4s), ptr_array(dfeats), ptr_array(dm3s), ptr_array(dzb4s),
         _ints(bs), len(as_), _stream(), label="hpp_bwd[bf16]", work=_hbm("hpp_bwd_kernel<2>", sum(bs) * 256 * 128 * 4.0 * 5))
    return dm3s, dzb4s
