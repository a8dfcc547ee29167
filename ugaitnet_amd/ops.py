"""Thin host wrappers over the C ABI: torch tensors are used as HBM buffers only (allocation + pointers).

Every function enqueues HIP kernels on torch's current stream and returns device tensors.  No arithmetic
happens in torch here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, ptr_array

F32 = torch.float32
U8 = torch.uint8


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype=F32):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.is_contiguous(), t.dtype)
    return t


class Workspace:
    """Grow-only scratch buffers for the weight-gradient partial slabs."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device):
        # one buffer per (device, stream): launches on different streams may run concurrently
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[key] = torch.empty(max(int(nbytes), 1), dtype=U8, device=device)
        return buf


_WS = Workspace()

# bench.py's serialised roofline pass (ugaitnet_amd._lib.PROFILE): every launch is filed under a label that names the device
# kernel it runs -- one label per template instantiation, so that the totals line up with rocprofv3's kernel-trace stats --
# together with what one launch does: algorithmic FLOPs (direct-convolution count), the FLOPs the matrix pipe
# executes (Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36 -> 16/36 of the algorithmic count) and, for the
# HBM-bound kernels, the algorithmic bytes.
def _conv_work(kind, hw, cin, cout, pooled, ns, bf16=False, wino=True, flags=0):
    n = int(sum(ns))
    flops = 2.0 * 9 * cin * cout * hw * hw * n
    bf = ", true" if bf16 else ""
    if kind == "fwd":
        epi = 1 if pooled else 0
        kern = ("wino_tall_kernel<%d, %d, 0, %d, 0%s>" % (cin, hw, epi, bf) if cout == 32
                else "wino_kernel<%d, %d, %d, 0, %d, 0%s>" % (cin, cout, hw, epi, bf))
    elif kind == "dgrad":
        kern = ("wino_tall_kernel<%d, %d, %d, 2, %d%s>" % (cout, hw, int(pooled), flags, bf) if cin == 32
                else "wino_kernel<%d, %d, %d, %d, 2, %d%s>" % (cout, cin, hw, int(pooled), flags, bf))
    else:
        kern = "wgrad_wino_kernel<%d, %d, %d, %d, %d%s>" % (cin, cout, hw, 64 if cout >= 64 else 32, int(pooled), bf)
    if not wino:
        kern = "conv3x3_kernel / wgrad3x3_kernel (direct)"
    label = "conv3x3_%s[%d->%d @%dx%d%s%s] %s" % (kind, cin, cout, hw, hw, " pooled" if pooled else "", " bf16" if bf16 else "", kern)
    return label, dict(flops=flops, mfma_flops=flops * (16.0 / 36.0 if wino else 1.0), bytes=None, kernel=kern, bound="mfma",
                       images=n, dtype="bf16" if bf16 else "f32")


def _no_bf16w(bf16):
    """The bf16-operand Winograd kernels on fp32 tensors ('bf16w', rounds 1-4) are retired: conv_precision='bf16' is the configs[4] path."""
    if bf16:
        raise ValueError("the bf16-operand Winograd kernels were retired in round 5; use conv_precision='bf16' (ugaitnet_amd.bf16)")
    return ""


def _hbm_work(kernel, nbytes):
    return dict(flops=None, mfma_flops=None, bytes=float(nbytes), kernel=kernel, bound="hbm")


def conv5x5_in_fwd(x, w, out=None, sign=None, x3=False):
    """sign (optional int32 [n,64,64]): receives one bit per output element, (a1 > 0), for conv5x5_in_wgrad.
    x3: multiply in the x3 arithmetic (three-way bf16 split, six products on the bf16 matrix pipe: csrc/x3_common.h) -- the form the
    default fp32-tensor set runs; False: v_mfma_f32_32x32x2_f32."""
    n, cin = x.shape[0], x.shape[3]
    _chk(x), _chk(w)
    out = torch.empty((n, 64, 64, 32), dtype=F32, device=x.device) if out is None else out
    if sign is not None:
        _chk(sign, torch.int32)
    call("ugn_x3_conv5x5_in_fwd" if x3 else "ugn_conv5x5_in_fwd", ptr(x), ptr(w), ptr(out), ptr(sign), n, cin, _stream(),
         label="conv5x5_fwd[cin=%d%s]" % (cin, " x3" if x3 else ""),
         work=_hbm_work("conv5x5_fwd_kernel<%d, %s, %d>" % (cin, "true" if sign is not None else "false", 3 if x3 else 0),
                        n * (3600.0 * cin * 4 + 4096 * 32 * 4 + (4096 * 4 if sign is not None else 0))))
    return out


def conv5x5_in_wgrad(x, dz1, dw=None, sign=None, x3=False):
    """sign given: dz1 is dL/da1 and the LeakyReLU' factor of a1 is applied inside, from the bits of conv5x5_in_fwd.
    x3: as conv5x5_in_fwd."""
    n, cin = x.shape[0], x.shape[3]
    _chk(x), _chk(dz1)
    dw = torch.empty((5, 5, cin, 32), dtype=F32, device=x.device) if dw is None else dw
    nbytes = _lib.load().ugn_conv5x5_in_wgrad_ws(n, cin)
    ws = _WS.get(nbytes, x.device)
    if sign is not None:
        _chk(sign, torch.int32)
    call("ugn_x3_conv5x5_in_wgrad" if x3 else "ugn_conv5x5_in_wgrad", ptr(x), ptr(dz1), ptr(sign), ptr(dw), n, cin, ptr(ws), ws.numel(),
         _stream(), label="conv5x5_wgrad[cin=%d%s]" % (cin, " x3" if x3 else ""),
         work=_hbm_work("conv5x5_wgrad_kernel<%d, %s, %d>" % (cin, "true" if sign is not None else "false", 4 if x3 else 0),
                        n * (3600.0 * cin * 4 + 4096 * 32 * 4 + (4096 * 4 if sign is not None else 0))))
    return dw


def pack3x3(w, out=None):
    _chk(w)
    cin, cout = w.shape[2], w.shape[3]
    out = torch.empty((9, cout, cin), dtype=F32, device=w.device) if out is None else out
    call("ugn_pack3x3", ptr(w), ptr(out), cin, cout, _stream())
    return out


def conv3x3_fwd(x, wp, pool, out=None, idx=None):
    """x [n,hw,hw,cin], wp packed [9,cout,cin] -> LeakyReLU(conv) ([n,hw,hw,cout]) or pooled + argmax index."""
    _chk(x), _chk(wp)
    n, hw, cin = x.shape[0], x.shape[1], x.shape[3]
    cout = wp.shape[1]
    ho = hw // 2 if pool else hw
    out = torch.empty((n, ho, ho, cout), dtype=F32, device=x.device) if out is None else out
    if pool and idx is None:
        idx = torch.empty((n, ho, ho, cout), dtype=U8, device=x.device)
    label, work = _conv_work("fwd", hw, cin, cout, pool, [n], wino=False)
    call("ugn_conv3x3_fwd", ptr(x), ptr(wp), ptr(out), ptr(idx) if pool else None, n, hw, cin, cout,
         int(bool(pool)), _stream(), label=label, work=work)
    return (out, idx) if pool else out


def conv3x3_dgrad(dz, w, hw, dz_idx=None, act=None, addend=None, out=None, raw_out=None):
    """w HWIO [3,3,cin,cout]; dz [n,hw,hw,cout] or pooled [n,hw/2,hw/2,cout] with dz_idx."""
    _chk(dz), _chk(w)
    n, cin, cout = dz.shape[0], w.shape[2], w.shape[3]
    out = torch.empty((n, hw, hw, cin), dtype=F32, device=dz.device) if out is None else out
    label, work = _conv_work("dgrad", hw, cin, cout, dz_idx is not None, [n], wino=False)
    call("ugn_conv3x3_dgrad", ptr(dz), ptr(dz_idx), ptr(w), ptr(act), ptr(addend), ptr(out), ptr(raw_out), n, hw, cin,
         cout, _stream(), label=label, work=work)
    return out


def _epi_flags(act, addend, raw_out):
    """The EFLAGS template argument of the data-gradient kernels (which epilogue operands are present)."""
    return (1 if act is not None else 0) | (2 if addend is not None else 0) | (4 if raw_out is not None else 0)


def _pack_mode(dgrad, pooled_dz, bf16=False):
    # 0: forward; 1: data gradient of a full-resolution dz; 3: data gradient of a pooled dz (+ argmax); + 4: bf16 elements
    return ((1 | (2 if pooled_dz else 0)) if dgrad else 0) | (4 if bf16 else 0)


def wino_pack(w, dgrad, out=None, pooled_dz=False, bf16=False):
    """HWIO [3,3,cin,cout] -> transformed filters G g G^T in the Winograd kernels' streaming order (16*cin*cout floats).
    `pooled_dz`: the data gradient will be called with dz_idx (the layer is followed by MaxPool).  `bf16`: elements rounded
    to bf16 (first half of the buffer), for the *_bf16 convolution entry points."""
    _chk(w)
    cin, cout = w.shape[2], w.shape[3]
    out = torch.empty((16 * cin * cout,), dtype=F32, device=w.device) if out is None else out
    call("ugn_wino_pack", ptr(w), ptr(out), cin, cout, _pack_mode(dgrad, pooled_dz, bf16), _stream())
    return out


def wino_pack_multi(jobs, bf16=False):
    """jobs: list of (w HWIO tensor, u_packed tensor, dgrad flag, pooled_dz flag); one launch for up to 64 of them."""
    for k in range(0, len(jobs), 64):
        part = jobs[k:k + 64]
        n = len(part)
        ws = ptr_array([j[0] for j in part])
        us = ptr_array([j[1] for j in part])
        cin = (C.c_int * n)(*[j[0].shape[2] for j in part])
        cout = (C.c_int * n)(*[j[0].shape[3] for j in part])
        dg = (C.c_int * n)(*[_pack_mode(j[2], j[3], bf16) for j in part])
        call("ugn_wino_pack_multi", ws, us, cin, cout, dg, n, _stream())


def conv3x3_fwd_wino(x, upk, cout, pool, out=None, idx=None, bf16=False):
    _chk(x), _chk(upk)
    n, hw, cin = x.shape[0], x.shape[1], x.shape[3]
    ho = hw // 2 if pool else hw
    out = torch.empty((n, ho, ho, cout), dtype=F32, device=x.device) if out is None else out
    if pool and idx is None:
        idx = torch.empty((n, ho, ho, cout), dtype=U8, device=x.device)
    label, work = _conv_work("fwd", hw, cin, cout, pool, [n], bf16)
    call("ugn_conv3x3_fwd_wino" + _no_bf16w(bf16), ptr(x), ptr(upk), ptr(out), ptr(idx) if pool else None, n, hw, cin, cout,
         int(bool(pool)), _stream(), label=label, work=work)
    return (out, idx) if pool else out


def conv3x3_dgrad_wino(dz, upk, hw, cin, cout, dz_idx=None, act=None, addend=None, out=None, raw_out=None, bf16=False):
    _chk(dz), _chk(upk)
    n = dz.shape[0]
    out = torch.empty((n, hw, hw, cin), dtype=F32, device=dz.device) if out is None else out
    label, work = _conv_work("dgrad", hw, cin, cout, dz_idx is not None, [n], bf16, flags=_epi_flags(act, addend, raw_out))
    call("ugn_conv3x3_dgrad_wino" + _no_bf16w(bf16), ptr(dz), ptr(dz_idx), ptr(upk), ptr(act), ptr(addend), ptr(out), ptr(raw_out), n, hw,
         cin, cout, _stream(), label=label, work=work)
    return out


def conv3x3_wgrad(x, dz, cout, dz_idx=None, dw=None):
    _chk(x), _chk(dz)
    n, hw, cin = x.shape[0], x.shape[1], x.shape[3]
    dw = torch.empty((3, 3, cin, cout), dtype=F32, device=x.device) if dw is None else dw
    nbytes = _lib.load().ugn_conv3x3_wgrad_ws(n, hw, cin, cout)
    if nbytes == 0:
        raise ValueError("conv3x3_wgrad: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _WS.get(nbytes, x.device)
    label, work = _conv_work("wgrad", hw, cin, cout, dz_idx is not None, [n], wino=False)
    call("ugn_conv3x3_wgrad", ptr(x), ptr(dz), ptr(dz_idx), ptr(dw), n, hw, cin, cout, ptr(ws), ws.numel(), _stream(),
         label=label, work=work)
    return dw


def _opt_ptr_array(pair):
    return None if pair is None or pair[0] is None else ptr_array(list(pair))


def conv3x3_fwd_wino_pair(xs, upks, cout, pool, outs, idxs=None, bf16=False):
    """Two forward convolutions of one shape (different inputs / filters / image counts) in a single launch."""
    for t in xs + upks + outs:
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    assert xs[1].shape[1:] == xs[0].shape[1:] and (not pool or idxs is not None)
    ns = (C.c_int * 2)(xs[0].shape[0], xs[1].shape[0])
    label, work = _conv_work("fwd", hw, cin, cout, pool, list(ns), bf16)
    call("ugn_conv3x3_fwd_wino_pair" + _no_bf16w(bf16), ptr_array(xs), ptr_array(upks), ptr_array(outs), _opt_ptr_array(idxs if pool else None),
         ns, hw, cin, cout, int(bool(pool)), _stream(), label=label, work=work)
    return (outs, idxs) if pool else outs


def conv3x3_dgrad_wino_pair(dzs, upks, hw, cin, cout, outs, dz_idxs=None, acts=None, addends=None, raw_outs=None, bf16=False):
    """Two data gradients of one shape in a single launch; the optional operands are given for both jobs or neither."""
    for t in dzs + upks + outs:
        _chk(t)
    ns = (C.c_int * 2)(dzs[0].shape[0], dzs[1].shape[0])
    flags = _epi_flags(acts and acts[0], addends and addends[0], raw_outs and raw_outs[0])
    label, work = _conv_work("dgrad", hw, cin, cout, bool(dz_idxs) and dz_idxs[0] is not None, list(ns), bf16, flags=flags)
    call("ugn_conv3x3_dgrad_wino_pair" + _no_bf16w(bf16), ptr_array(dzs), _opt_ptr_array(dz_idxs), ptr_array(upks), _opt_ptr_array(acts),
         _opt_ptr_array(addends), ptr_array(outs), _opt_ptr_array(raw_outs), ns, hw, cin, cout, _stream(), label=label, work=work)
    return outs


def conv3x3_wgrad_wino_pair(xs, dzs, cout, dws, dz_idxs=None, bf16=False):
    """Two weight gradients of one shape in a single launch."""
    for t in xs + dzs + dws:
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    ns = (C.c_int * 2)(xs[0].shape[0], xs[1].shape[0])
    nbytes = _lib.load().ugn_conv3x3_wgrad_wino_ws(ns[0] + ns[1], hw, cin, cout)
    if nbytes == 0:
        raise ValueError("conv3x3_wgrad_wino_pair: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _WS.get(nbytes, xs[0].device)
    label, work = _conv_work("wgrad", hw, cin, cout, bool(dz_idxs) and dz_idxs[0] is not None, list(ns), bf16)
    call("ugn_conv3x3_wgrad_wino_pair" + _no_bf16w(bf16), ptr_array(xs), ptr_array(dzs), _opt_ptr_array(dz_idxs), ptr_array(dws), ns, hw, cin, cout,
         ptr(ws), ws.numel(), _stream(), label=label, work=work)
    return dws


MAX_JOBS = 6   # jobs per multi launch (wino_common.h kMaxJobs): 3 modalities x (frame-level layer + set-level twin)


def _int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def conv3x3_fwd_wino_multi(xs, upks, cout, pool, outs, idxs=None, bf16=False):
    """Up to 6 forward convolutions of one shape (own inputs / filters / image counts) in a single launch."""
    assert 1 <= len(xs) <= MAX_JOBS and len(upks) == len(outs) == len(xs) and (not pool or idxs is not None)
    for t in list(xs) + list(upks) + list(outs):
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    assert all(x.shape[1:] == xs[0].shape[1:] for x in xs)
    ns = [x.shape[0] for x in xs]
    label, work = _conv_work("fwd", hw, cin, cout, pool, ns, bf16)
    call("ugn_conv3x3_fwd_wino_multi", ptr_array(xs), ptr_array(upks), ptr_array(outs), ptr_array(idxs) if pool else None,
         _int_array(ns), len(xs), hw, cin, cout, int(bool(pool)), int(bool(bf16)), _stream(), label=label, work=work)
    return (outs, idxs) if pool else outs


def conv3x3_dgrad_wino_multi(dzs, upks, hw, cin, cout, outs, dz_idxs=None, acts=None, addends=None, raw_outs=None, bf16=False):
    """Up to 6 data gradients of one shape in a single launch; optional operands for all jobs or none."""
    assert 1 <= len(dzs) <= MAX_JOBS and len(upks) == len(outs) == len(dzs)
    for t in list(dzs) + list(upks) + list(outs):
        _chk(t)
    ns = [d.shape[0] for d in dzs]
    flags = _epi_flags(acts and acts[0], addends and addends[0], raw_outs and raw_outs[0])
    label, work = _conv_work("dgrad", hw, cin, cout, bool(dz_idxs) and dz_idxs[0] is not None, ns, bf16, flags=flags)
    call("ugn_conv3x3_dgrad_wino_multi", ptr_array(dzs), _opt_ptr_array(dz_idxs), ptr_array(upks), _opt_ptr_array(acts),
         _opt_ptr_array(addends), ptr_array(outs), _opt_ptr_array(raw_outs), _int_array(ns), len(dzs), hw, cin, cout,
         int(bool(bf16)), _stream(), label=label, work=work)
    return outs


def conv3x3_wgrad_wino_multi(xs, dzs, cout, dws, dz_idxs=None, bf16=False):
    """Up to 6 weight gradients of one shape in a single launch."""
    assert 1 <= len(xs) <= MAX_JOBS and len(dzs) == len(dws) == len(xs)
    for t in list(xs) + list(dzs) + list(dws):
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    ns = [x.shape[0] for x in xs]
    nbytes = _lib.load().ugn_conv3x3_wgrad_wino_ws(sum(ns), hw, cin, cout)
    if nbytes == 0:
        raise ValueError("conv3x3_wgrad_wino_multi: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _WS.get(nbytes, xs[0].device)
    label, work = _conv_work("wgrad", hw, cin, cout, bool(dz_idxs) and dz_idxs[0] is not None, ns, bf16)
    call("ugn_conv3x3_wgrad_wino_multi", ptr_array(xs), ptr_array(dzs), _opt_ptr_array(dz_idxs), ptr_array(dws), _int_array(ns),
         len(xs), hw, cin, cout, ptr(ws), ws.numel(), int(bool(bf16)), _stream(), label=label, work=work)
    return dws


def scale_(x, factor):
    """x *= factor in place (a HIP kernel, not torch arithmetic)."""
    _chk(x)
    call("ugn_scale", ptr(x), float(factor), x.numel(), _stream())
    return x


def conv3x3_wgrad_wino(x, dz, cout, dz_idx=None, dw=None, bf16=False):
    """Winograd F(2x2,3x3) weight gradient; same contract as conv3x3_wgrad."""
    _chk(x), _chk(dz)
    n, hw, cin = x.shape[0], x.shape[1], x.shape[3]
    dw = torch.empty((3, 3, cin, cout), dtype=F32, device=x.device) if dw is None else dw
    nbytes = _lib.load().ugn_conv3x3_wgrad_wino_ws(n, hw, cin, cout)
    if nbytes == 0:
        raise ValueError("conv3x3_wgrad_wino: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _WS.get(nbytes, x.device)
    label, work = _conv_work("wgrad", hw, cin, cout, dz_idx is not None, [n], bf16)
    call("ugn_conv3x3_wgrad_wino" + _no_bf16w(bf16), ptr(x), ptr(dz), ptr(dz_idx), ptr(dw), n, hw, cin, cout, ptr(ws), ws.numel(), _stream(),
         label=label, work=work)
    return dw


def setmax_fwd(p, b, l, addend=None, m=None, sum_out=None):
    _chk(p)
    s = p.numel() // (b * l)
    shape = tuple(p.shape[1:])
    m = torch.empty((b,) + shape, dtype=F32, device=p.device) if m is None else m
    if addend is not None and sum_out is None:
        sum_out = torch.empty_like(m)
    call("ugn_setmax_fwd", ptr(p), ptr(addend), ptr(m), ptr(sum_out), b, l, s, _stream(), label="setmax_fwd[%d x %d x %d]" % (b, l, s),
         work=_hbm_work("setmax_fwd_kernel", 4.0 * s * b * (l + 1 + (2 if addend is not None else 0))))
    return (m, sum_out) if addend is not None else m


def setmax_fwd_cnt(p, b, l, addend=None, m=None, sum_out=None, cnt=None):
    """setmax_fwd that also returns the number of frames holding each maximum (for the routed data gradient)."""
    _chk(p)
    s = p.numel() // (b * l)
    shape = tuple(p.shape[1:])
    m = torch.empty((b,) + shape, dtype=F32, device=p.device) if m is None else m
    cnt = torch.empty((b,) + shape, dtype=F32, device=p.device) if cnt is None else cnt
    if addend is not None and sum_out is None:
        sum_out = torch.empty_like(m)
    call("ugn_setmax_fwd_cnt", ptr(p), ptr(addend), ptr(m), ptr(sum_out), ptr(cnt), b, l, s, _stream())
    return (m, sum_out, cnt) if addend is not None else (m, cnt)


def lrelu_bwd(g, act, out=None):
    """g * LeakyReLU'(act), act being the LeakyReLU output."""
    _chk(g), _chk(act)
    out = torch.empty_like(g) if out is None else out
    call("ugn_lrelu_bwd", ptr(g), ptr(act), ptr(out), g.numel(), _stream())
    return out


def div(a, b, out=None):
    _chk(a), _chk(b)
    out = torch.empty_like(a) if out is None else out
    call("ugn_div", ptr(a), ptr(b), ptr(out), a.numel(), _stream())
    return out


def conv3x3_dgrad_wino_routed(dz, upk, hw, cin, cout, act, smax_m, smax_g, frames, out=None):
    """Data gradient + set-max gradient of the layer's output formed in the epilogue + LeakyReLU' (see the header)."""
    for t in (dz, upk, act, smax_m, smax_g):
        _chk(t)
    n = dz.shape[0]
    out = torch.empty((n, hw, hw, cin), dtype=F32, device=dz.device) if out is None else out
    call("ugn_conv3x3_dgrad_wino_routed", ptr(dz), ptr(upk), ptr(act), ptr(smax_m), ptr(smax_g), int(frames), ptr(out), n, hw, cin,
         cout, _stream())
    return out


def setmax_bwd(p, dm, b, l, apply_lrelu, out=None, addend=None):
    """addend (may be `out` itself): a second gradient path into p, added before the LeakyReLU' factor."""
    _chk(p), _chk(dm)
    s = p.numel() // (b * l)
    out = torch.empty_like(p) if out is None else out
    if addend is not None:
        _chk(addend)
    call("ugn_setmax_bwd", ptr(p), ptr(dm), ptr(addend), ptr(out), b, l, s, int(bool(apply_lrelu)), _stream(),
         label="setmax_bwd[%d x %d x %d%s]" % (b, l, s, " +addend" if addend is not None else ""), work=_hbm_work("setmax_bwd_kernel", 4.0 * s * b * (2 * l + 1 + (l if addend is not None else 0))))
    return out


# ---- the per-branch pooling / FC steps for several modality branches in one launch (lists of equal length; the tensors of
# a list share their trailing shape, the clip counts may differ)
def setmax_fwd_multi(ps, bs, l, ms, addends=None, sum_outs=None):
    for t in list(ps) + list(ms):
        _chk(t)
    s = ps[0].numel() // (bs[0] * l)
    nb = sum(4.0 * s * b * (l + 1 + (2 if addends is not None else 0)) for b in bs)
    call("ugn_setmax_fwd_multi", ptr_array(ps), _opt_ptr_array(addends), ptr_array(ms), _opt_ptr_array(sum_outs), _int_array(bs),
         len(ps), l, s, _stream(), label="setmax_fwd[%s x %d x %d]" % ("+".join(str(b) for b in bs), l, s),
         work=_hbm_work("setmax_fwd_kernel", nb))
    return (ms, sum_outs) if addends is not None else ms


def setmax_bwd_multi(ps, dms, bs, l, apply_lrelu, outs, addends=None):
    for t in list(ps) + list(dms) + list(outs):
        _chk(t)
    s = ps[0].numel() // (bs[0] * l)
    nb = sum(4.0 * s * b * (2 * l + 1 + (l if addends is not None else 0)) for b in bs)
    call("ugn_setmax_bwd_multi", ptr_array(ps), ptr_array(dms), _opt_ptr_array(addends), ptr_array(outs), _int_array(bs), len(ps),
         l, s, int(bool(apply_lrelu)), _stream(),
         label="setmax_bwd[%s x %d x %d%s]" % ("+".join(str(b) for b in bs), l, s, " +addend" if addends is not None else ""),
         work=_hbm_work("setmax_bwd_kernel", nb))
    return outs


def setmax_fwd_routed_multi(ps, bs, l, ms, routes, addends=None, sum_outs=None):
    """setmax_fwd_multi that also writes the routing words (int32 [b, s/4, 2, 4]) the routed gradient reads instead of the frames."""
    for t in list(ps) + list(ms):
        _chk(t)
    for r in routes:
        _chk(r, torch.int32)
    s = ps[0].numel() // (bs[0] * l)
    nb = sum(4.0 * s * b * (l + 3 + (2 if addends is not None else 0)) for b in bs)
    call("ugn_setmax_fwd_routed_multi", ptr_array(ps), _opt_ptr_array(addends), ptr_array(ms), _opt_ptr_array(sum_outs), ptr_array(routes),
         _int_array(bs), len(ps), l, s, _stream(), label="setmax_fwd_routed[%s x %d x %d]" % ("+".join(str(b) for b in bs), l, s),
         work=_hbm_work("setmax_fwd_routed_kernel", nb))
    return (ms, sum_outs) if addends is not None else ms


def setmax_bwd_routed_multi(routes, dms, bs, l, apply_lrelu, outs, addends=None):
    """setmax_bwd_multi from the routing words: 4 l (+ 4 l with an addend) + 12 bytes per set element instead of 8 l (+ 4 l) + 4."""
    for t in list(dms) + list(outs):
        _chk(t)
    s = outs[0].numel() // (bs[0] * l)
    nb = sum(4.0 * s * b * (l + 3 + (l if addends is not None else 0)) for b in bs)
    call("ugn_setmax_bwd_routed_multi", ptr_array(routes), ptr_array(dms), _opt_ptr_array(addends), ptr_array(outs), _int_array(bs), len(outs),
         l, s, int(bool(apply_lrelu)), _stream(),
         label="setmax_bwd_routed[%s x %d x %d%s]" % ("+".join(str(b) for b in bs), l, s, " +addend" if addends is not None else ""),
         work=_hbm_work("setmax_bwd_routed_kernel", nb))
    return outs


def lrelu_bwd_multi(gs, acts, outs):
    for t in list(gs) + list(acts) + list(outs):
        _chk(t)
    ns = (C.c_size_t * len(gs))(*[g.numel() for g in gs])
    call("ugn_lrelu_bwd_multi", ptr_array(gs), ptr_array(acts), ptr_array(outs), ns, len(gs), _stream())
    return outs


def hpp_fwd_multi(as_, s3s, feats):
    bs = [a.shape[0] for a in as_]
    call("ugn_hpp_fwd_multi", ptr_array([_chk(a) for a in as_]), ptr_array([_chk(t) for t in s3s]), ptr_array(feats), _int_array(bs),
         len(as_), _stream())
    return feats


def hpp_bwd_multi(as_, s3s, b4s, dfeats, dm3s, dzb4s):
    bs = [a.shape[0] for a in as_]
    call("ugn_hpp_bwd_multi", ptr_array([_chk(a) for a in as_]), ptr_array([_chk(t) for t in s3s]), ptr_array([_chk(t) for t in b4s]),
         ptr_array([_chk(t) for t in dfeats]), ptr_array(dm3s), ptr_array(dzb4s), _int_array(bs), len(as_), _stream())
    return dm3s, dzb4s


def binfc_fwd_multi(feats, ws, outs):
    bs = [f.shape[1] for f in feats]
    call("ugn_binfc_fwd_multi", ptr_array([_chk(f) for f in feats]), ptr_array([_chk(w) for w in ws]), ptr_array(outs),
         _int_array(bs), len(feats), _stream())
    return outs


def binfc_bwd_multi(feats, ws, douts, dws, dfeats, parts=3):
    """parts: 1 = the weight gradients only, 2 = the feature gradients only (what HPP backward waits for), 3 = both."""
    bs = [f.shape[1] for f in feats]
    call("ugn_binfc_bwd_parts_multi", ptr_array([_chk(f) for f in feats]), ptr_array([_chk(w) for w in ws]),
         ptr_array([_chk(d) for d in douts]), ptr_array(dws), ptr_array(dfeats), _int_array(bs), len(feats), int(parts), _stream(),
         label="ugn_binfc_bwd_multi" + {1: "[dW]", 2: "[dfeat]", 3: ""}[int(parts)])
    return dws, dfeats


def hpp_fwd(a, s3, feat=None):
    b = a.shape[0]
    feat = torch.empty((62, b, 128), dtype=F32, device=a.device) if feat is None else feat
    call("ugn_hpp_fwd", ptr(_chk(a)), ptr(_chk(s3)), ptr(feat), b, _stream())
    return feat


def hpp_bwd(a, s3, b4, dfeat, dm3=None, dzb4=None):
    b = a.shape[0]
    dm3 = torch.empty_like(a) if dm3 is None else dm3
    dzb4 = torch.empty_like(a) if dzb4 is None else dzb4
    call("ugn_hpp_bwd", ptr(_chk(a)), ptr(_chk(s3)), ptr(_chk(b4)), ptr(_chk(dfeat)), ptr(dm3), ptr(dzb4), b, _stream())
    return dm3, dzb4


def binfc_fwd(feat, w, out=None):
    b = feat.shape[1]
    out = torch.empty((62, b, 256), dtype=F32, device=feat.device) if out is None else out
    call("ugn_binfc_fwd", ptr(_chk(feat)), ptr(_chk(w)), ptr(out), b, _stream())
    return out


def binfc_bwd(feat, w, dout, dw=None, dfeat=None):
    b = feat.shape[1]
    dw = torch.empty_like(w) if dw is None else dw
    dfeat = torch.empty_like(feat) if dfeat is None else dfeat
    call("ugn_binfc_bwd", ptr(_chk(feat)), ptr(_chk(w)), ptr(_chk(dout)), ptr(dw), ptr(dfeat), b, _stream())
    return dw, dfeat


def gate_fuse_fwd(outs, uses, mode, fused=None, sel=None):
    b = outs[0].shape[1]
    fused = torch.empty_like(outs[0]) if fused is None else fused
    sel = torch.empty(outs[0].shape, dtype=U8, device=outs[0].device) if sel is None else sel
    call("ugn_gate_fuse_fwd", ptr_array(outs), ptr_array(uses), len(outs), _lib.FUSE_MODES[mode], ptr(fused), ptr(sel), b,
         _stream())
    return fused, sel


def gate_fuse_bwd(dfused, sel, uses, mode, douts=None):
    b = dfused.shape[1]
    douts = [torch.empty_like(dfused) for _ in uses] if douts is None else douts
    call("ugn_gate_fuse_bwd", ptr(_chk(dfused)), ptr(sel), ptr_array(uses), ptr_array(douts), len(uses),
         _lib.FUSE_MODES[mode], b, _stream())
    return douts


GATE_NORM_MAXB = 32


def gate_norm_fwd(outs, uses, mode, fused, sel, sig):
    """gate_fuse_fwd + l2norm_batch_fwd in one launch (at most GATE_NORM_MAXB clips): the same three outputs."""
    call("ugn_gate_norm_fwd", ptr_array(outs), ptr_array(uses), len(outs), _lib.FUSE_MODES[mode], ptr(fused), ptr(sel), ptr(sig),
         outs[0].shape[1], _stream())
    return fused, sel, sig


def gate_norm_bwd(f, sig, dsig, sel, uses, mode, douts):
    """l2norm_batch_bwd + gate_fuse_bwd in one launch (at most GATE_NORM_MAXB clips): the same douts."""
    call("ugn_gate_norm_bwd", ptr(_chk(f)), ptr(_chk(sig)), ptr(_chk(dsig)), ptr(sel), ptr_array(uses), ptr_array(douts), len(uses),
         _lib.FUSE_MODES[mode], f.shape[1], _stream())
    return douts


def l2norm_batch_fwd(f, sig=None):
    sig = torch.empty_like(f) if sig is None else sig
    call("ugn_l2norm_batch_fwd", ptr(_chk(f)), ptr(sig), f.shape[1], _stream())
    return sig


def l2norm_batch_bwd(f, sig, dsig, df=None):
    df = torch.empty_like(f) if df is None else df
    call("ugn_l2norm_batch_bwd", ptr(_chk(f)), ptr(_chk(sig)), ptr(_chk(dsig)), ptr(df), f.shape[1], _stream())
    return df


def head_fwd(sig, wc, bc, onehot, grad_scale, bufs=None):
    b, ncls = sig.shape[1], wc.shape[1]
    dev = sig.device
    if bufs is None:
        bufs = dict(part=torch.empty((248, b, ncls), dtype=F32, device=dev),
                    probs=torch.empty((b, ncls), dtype=F32, device=dev),
                    row_loss=torch.empty((b,), dtype=F32, device=dev),
                    dlogits=torch.empty((b, ncls), dtype=F32, device=dev),
                    hit=torch.empty((b,), dtype=F32, device=dev))
    call("ugn_head_fwd", ptr(_chk(sig)), ptr(_chk(wc)), ptr(_chk(bc)), ptr(_chk(onehot)), ptr(bufs["part"]),
         ptr(bufs["probs"]), ptr(bufs["row_loss"]), ptr(bufs["dlogits"]), ptr(bufs["hit"]), float(grad_scale), b, ncls,
         _stream())
    return bufs


def head_bwd(sig, wc, dlogits, dsig, accumulate, dwc=None, dbc=None):
    b, ncls = sig.shape[1], wc.shape[1]
    dwc = torch.empty_like(wc) if dwc is None else dwc
    dbc = torch.empty((ncls,), dtype=F32, device=sig.device) if dbc is None else dbc
    call("ugn_head_bwd", ptr(_chk(sig)), ptr(_chk(wc)), ptr(_chk(dlogits)), ptr(dwc), ptr(dbc), ptr(dsig),
         int(bool(accumulate)), b, ncls, _stream())
    return dwc, dbc


def triplet_indices(labels):
    """labels: 1-D integer array (host).  Returns (hp, hn, kp, kn) with hp/hn int32 numpy arrays."""
    lab = np.ascontiguousarray(np.asarray(labels).reshape(-1), dtype=np.int32)
    m = lab.shape[0]
    hp = np.empty(m * m, dtype=np.int32)
    hn = np.empty(m * m, dtype=np.int32)
    kp, kn = C.c_int(0), C.c_int(0)
    call("ugn_triplet_indices_host", lab.ctypes.data_as(C.c_void_p), m, hp.ctypes.data_as(C.c_void_p),
         hn.ctypes.data_as(C.c_void_p), C.byref(kp), C.byref(kn))
    return hp[:m * kp.value].copy(), hn[:m * kn.value].copy(), kp.value, kn.value


def triplet_fwd_bwd(sig, hp, hn, kp, kn, margin, grad_scale, bin_loss=None, bin_num=None, dsig=None):
    m = sig.shape[1]
    dev = sig.device
    bin_loss = torch.empty((62,), dtype=F32, device=dev) if bin_loss is None else bin_loss
    bin_num = torch.empty((62,), dtype=F32, device=dev) if bin_num is None else bin_num
    dsig = torch.empty_like(sig) if dsig is None else dsig
    call("ugn_triplet_fwd_bwd", ptr(_chk(sig)), ptr(hp), ptr(hn), kp, kn, float(margin), ptr(bin_loss), ptr(bin_num),
         ptr(dsig), float(grad_scale), m, _stream())
    return bin_loss, bin_num, dsig


def triplet_hard_fwd_bwd(sig, labels_dev, margin, grad_scale, bin_loss=None, bin_num=None, dsig=None):
    """Batch-hard triplet loss per bin (tfa.losses.TripletHardLoss semantics); labels_dev: int32 [m] on the device."""
    m = sig.shape[1]
    dev = sig.device
    bin_loss = torch.empty((62,), dtype=F32, device=dev) if bin_loss is None else bin_loss
    bin_num = torch.empty((62,), dtype=F32, device=dev) if bin_num is None else bin_num
    dsig = torch.empty_like(sig) if dsig is None else dsig
    call("ugn_triplet_hard_fwd_bwd", ptr(_chk(sig)), ptr(_chk(labels_dev, torch.int32)), float(margin), ptr(bin_loss), ptr(bin_num),
         ptr(dsig), float(grad_scale), m, _stream())
    return bin_loss, bin_num, dsig


def adam_step(p, g, m, v, lr_t, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0):
    call("ugn_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr_t), float(b1), float(b2), float(eps),
         float(grad_scale), _stream(), work=_hbm_work("adam_kernel", 28.0 * p.numel()))


def adam_step_dev(p, g, m, v, lr_t_dev, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0):
    """adam_step with lr_t in device memory (a 1-element fp32 tensor): capturable in a hipGraph."""
    call("ugn_adam_step_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(_chk(lr_t_dev)), float(b1), float(b2), float(eps),
         float(grad_scale), _stream())


def gather_rows(src, idx, axis, out=None):
    """out = src.index_select(axis, idx) for axis 0 ([B, ...] clip tensors) or 1 ([62, B, 256] features), as a HIP row copy
    (ugn_gather_rows): the mask-skipping step's dense sub-batches.  idx: int64 device tensor."""
    _chk(src)
    outer = 1 if axis == 0 else src.shape[0]
    rows = src.shape[axis]
    row = int(np.prod(src.shape[axis + 1:]))
    shape = list(src.shape)
    shape[axis] = idx.numel()
    out = torch.empty(shape, dtype=F32, device=src.device) if out is None else out
    call("ugn_gather_rows", ptr(src), ptr(idx), ptr(out), outer, rows, idx.numel(), row, _stream())
    return out


def scatter_rows(src, idx, axis, dst):
    """dst.index_copy_(axis, idx, src) as a HIP row copy (ugn_scatter_rows); rows of dst that idx does not name are left alone."""
    _chk(src), _chk(dst)
    outer = 1 if axis == 0 else dst.shape[0]
    row = int(np.prod(dst.shape[axis + 1:]))
    call("ugn_scatter_rows", ptr(src), ptr(idx), ptr(dst), outer, dst.shape[axis], idx.numel(), row, _stream())
    return dst


def set_persistent_wgs(n):
    """Workgroups of the library's persistent launches (0 = default, one per CU; 8..256): the ONE process-wide launch setting,
    see include/ugaitnet_hip.h ugn_set_persistent_wgs.  Results do not depend on it."""
    _lib.check(_lib.load().ugn_set_persistent_wgs(int(n)), "ugn_set_persistent_wgs")


def get_persistent_wgs():
    """The persistent grid in force (8..256)."""
    return int(_lib.load().ugn_get_persistent_wgs())

