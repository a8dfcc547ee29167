"""H2 tensors (split-fp16 halves + block exponent) and the host wrappers of the kernels that consume them.

Format and rationale: ugaitnet_amd/csrc/mm_common.h.  An H2 tensor replaces an fp32 NHWC activation / gradient between the
3x3 layers of the encoder (reference nets/mj_uwyhNets_ba.py:431-462): same bytes, but the halves feed
v_mfma_f32_32x32x16_f16 directly.  torch holds the buffers; every arithmetic op is a C-ABI call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, ptr_array

I16 = torch.int16
I32 = torch.int32
U8 = torch.uint8
F32 = torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MetaPool:
    """All ugn_h2meta records of a model in ONE int32 buffer: the per-step reset is a single memset.
    Slots are handed out by name and live as long as the pool.

    ONE RECORD PER 256 BYTES.  A producer gathers `amax` with atomicMax (thousands per launch, all on one address, ~0.1 us
    each at the L2); a consumer -- or the same kernel, for its INPUT's record -- loads the neighbouring record.  Side by side
    in one cache line those loads queue behind the atomics: measured on MI355X, the 32 -> 64 forward launch of the C3 step took
    447 us with input and output records adjacent and 247 us with either of them moved away (tools/bench_step_kernels.py)."""
    STRIDE = 64      # int32 per record (the record itself is the first two)

    def __init__(self, device, capacity=1024):
        self.buf = torch.zeros((capacity, self.STRIDE), dtype=I32, device=device)
        self.slots = {}

    def slot(self, name):
        i = self.slots.get(name)
        if i is None:
            i = self.slots[name] = len(self.slots)
            if i >= self.buf.shape[0]:
                raise RuntimeError("MetaPool: out of slots")
        return self.buf[i, :2]

    def reset(self, keep=()):
        """Zero every record (amax gathers by atomicMax) except the named ones (filter metas survive the step)."""
        if not keep:
            self.buf.zero_()
            return
        saved = [(self.slots[k], self.buf[self.slots[k]].clone()) for k in keep if k in self.slots]
        self.buf.zero_()
        for i, v in saved:
            self.buf[i].copy_(v)


class H2Tensor:
    """data: int16 [n, h, w, 2, c] (f16 bit patterns, plane 0 = H, plane 1 = L); meta: int32 [2] = {e, bits(amax)}."""

    __slots__ = ("data", "meta")

    def __init__(self, data, meta):
        assert data.dtype == I16 and data.is_contiguous() and data.dim() == 5 and data.shape[3] == 2, (data.dtype, data.shape)
        assert meta.dtype == I32 and meta.numel() == 2
        self.data, self.meta = data, meta

    @property
    def shape(self):          # logical NHWC shape
        n, h, w, _, c = self.data.shape
        return (n, h, w, c)

    @property
    def device(self):
        return self.data.device

    @staticmethod
    def empty(shape, device, meta=None):
        n, h, w, c = shape
        meta = torch.zeros(2, dtype=I32, device=device) if meta is None else meta
        return H2Tensor(torch.empty((n, h, w, 2, c), dtype=I16, device=device), meta)

    def numpy(self):
        """Decode on the HOST (tests): float64 array of the true values."""
        e, amax = self.meta.cpu().numpy().tolist()
        d = self.data.cpu().numpy().view(np.float16).astype(np.float64)
        return (d[:, :, :, 0, :] + d[:, :, :, 1, :]) * 2.0 ** (-e)

    def true_amax(self):
        e, bits = self.meta.cpu().numpy().tolist()
        return float(np.array([bits], np.uint32).view(np.float32)[0]) * 2.0 ** (-e)


def encode(x, out=None):
    """fp32 NHWC tensor -> H2Tensor (three small kernels: max |x|, split, meta)."""
    assert x.is_cuda and x.dtype == F32 and x.is_contiguous() and x.dim() == 4
    n, h, w, c = x.shape
    out = H2Tensor.empty((n, h, w, c), x.device) if out is None else out
    call("ugn_h2_encode", ptr(x), ptr(out.data), ptr(out.meta), n * h * w, c, _stream())
    return out


def decode(t, out=None):
    n, h, w, c = t.shape
    out = torch.empty((n, h, w, c), dtype=F32, device=t.device) if out is None else out
    call("ugn_h2_decode", ptr(t.data), ptr(t.meta), ptr(out), n * h * w, c, _stream())
    return out


def set_persistent_wgs(n):
    """Workgroups of the persistent forward / data-gradient launches (0 = default, one per CU): ops.set_persistent_wgs."""
    from . import ops
    ops.set_persistent_wgs(n)


def absmax(x, meta):
    """meta <- {0, bits(max|x|)} (meta zero on entry)."""
    call("ugn_absmax", ptr(x), x.numel(), ptr(meta), _stream())
    return meta


def mm_pack_multi(jobs):
    """jobs: list of (w HWIO fp32 tensor, packed int16 tensor of 18*cin*cout elements, wmeta int32[4], flags): flags bit 0 = data
    gradient (taps mirrored, K = cout), bit 1 = the 32-column block layout whatever the shape's default kernel (dgrad32_wgrad5_multi)."""
    for k in range(0, len(jobs), 64):
        part = jobs[k:k + 64]
        n = len(part)
        call("ugn_mm_pack_multi", ptr_array([j[0] for j in part]), ptr_array([j[1] for j in part]), ptr_array([j[2] for j in part]),
             (C.c_int * n)(*[j[0].shape[2] for j in part]), (C.c_int * n)(*[j[0].shape[3] for j in part]),
             (C.c_int * n)(*[int(j[3]) & 3 for j in part]), n, _stream())


def mm_pack(w, dgrad, pk=None, wmeta=None):
    cin, cout = w.shape[2], w.shape[3]
    pk = torch.empty((18 * cin * cout,), dtype=I16, device=w.device) if pk is None else pk
    wmeta = torch.zeros(4, dtype=I32, device=w.device) if wmeta is None else wmeta      # ugn_wmeta: 16 bytes
    mm_pack_multi([(w, pk, wmeta, dgrad)])
    return pk, wmeta


def _mm_bytes(kind, hw, cin, cout, pooled, n, act=False):
    """Algorithmic HBM bytes of a 3x3 H2 launch: every tensor it must read or write once (H2 = 4 bytes per element, argmax maps 1,
    LeakyReLU' reads the H plane of the layer's input = 2; the packed filters, re-read from L2 by every workgroup, are not counted)."""
    hp = hw // 2
    if kind == "fwd":       # in [n,hw,hw,cin] -> out [n,ho,ho,cout] (+ argmax bytes)
        return n * (hw * hw * cin * 4.0 + (hp * hp * cout * 5.0 if pooled else hw * hw * cout * 4.0))
    dz = hp * hp * cout * 5.0 if pooled else hw * hw * cout * 4.0      # the gradient operand (pooled: + argmax bytes)
    if kind == "dgrad":     # dz -> dL/d(in) [n,hw,hw,cin] (* LeakyReLU'(act))
        return n * (dz + hw * hw * cin * 4.0 + (hw * hw * cin * 2.0 if act else 0.0))
    return n * (dz + hw * hw * cin * 4.0)      # wgrad: the layer's input + dz (the fp32 result is a few hundred KB)


def _mm_work(kind, hw, cin, cout, pooled, ns, act=False):
    n = int(sum(ns))
    flops = 2.0 * 9 * cin * cout * hw * hw * n
    # (the names rocprofv3 reports, csrc/conv3x3_mm.hip: the 32 -> 32 @64x64 layer runs the two-workgroups-per-CU kernels, launches
    #  with 128 output columns and the pooled 64 -> 64 data gradient conv_nr_kernel, the other un-pooled inputs conv_mm16_kernel)
    a2 = cin == 32 and cout == 32 and hw == 64 and pooled
    t16 = lambda kc, nc, epi: ("conv_nr_kernel<%d, %d, %d, %d, 0>" if nc >= 128 else "conv_mm16_kernel<%d, %d, %d, %d>") % (kc, nc, hw, epi)
    if kind == "fwd":
        kern = "conv_d2_kernel<32, 32, 64, 1, 0>" if a2 else t16(cin, cout, 1 if pooled else 0)
    elif a2:
        kern = "conv_d2_kernel<32, 32, 64, %d, 1>" % (3 if act else 2)
    else:
        kern = ("conv_nr_kernel<%d, %d, %d, %d, 1>" % (cout, cin, hw, 3 if act else 2) if pooled else t16(cout, cin, 3 if act else 2))
    label = "conv3x3_%s[%d->%d @%dx%d%s h2] %s" % (kind, cin, cout, hw, hw, " pooled" if pooled else "", kern)
    # three f16 MFMAs per fp32-equivalent product: the matrix pipe executes 3x the algorithmic FLOPs.  bound="roof": bench.py prices
    # the launch against BOTH roofs (executed matrix FLOPs, algorithmic bytes) and names the one whose floor is higher
    return label, dict(flops=flops, mfma_flops=3.0 * flops, bytes=_mm_bytes(kind, hw, cin, cout, pooled, n, act), kernel=kern,
                       bound="roof", images=n, dtype="f16x2")


def conv3x3_fwd_mm_multi(xs, wpks, wmetas, cout, pool, outs, idxs=None):
    """LeakyReLU(conv3x3(x)) (+ MaxPool + argmax) for up to 6 jobs of one shape in one launch.  xs / outs: H2Tensors."""
    n_, hw, _, cin = xs[0].shape
    ns = (C.c_int * len(xs))(*[x.shape[0] for x in xs])
    ho = hw // 2 if pool else hw
    for x, o in zip(xs, outs):
        assert x.shape[1:] == (hw, hw, cin) and o.shape == (x.shape[0], ho, ho, cout), (x.shape, o.shape)
    label, work = _mm_work("fwd", hw, cin, cout, pool, [x.shape[0] for x in xs])
    call("ugn_mm_conv3x3_fwd_multi", ptr_array([x.data for x in xs]), ptr_array([x.meta for x in xs]), ptr_array(wpks),
         ptr_array(wmetas), ptr_array([o.data for o in outs]), ptr_array(idxs) if pool else None,
         ptr_array([o.meta for o in outs]), ns, len(xs), hw, cin, cout, int(bool(pool)), _stream(), label=label, work=work)
    return (outs, idxs) if pool else outs


def conv3x3_dgrad_mm_multi(dzs, wpks, wmetas, hw, cin, cout, outs, dz_idxs=None, acts=None):
    """Data gradient of the layer cin -> cout at hw x hw for up to 6 jobs.  dzs: H2 gradients w.r.t. the layer's
    pre-activation ([n,hw,hw,cout]; with dz_idxs: pooled [n,hw/2,hw/2,cout] + argmax bytes).  acts: H2 inputs of the layer
    (out *= LeakyReLU'(act))."""
    ns = (C.c_int * len(dzs))(*[d.shape[0] for d in dzs])
    for d, o in zip(dzs, outs):
        assert o.shape == (d.shape[0], hw, hw, cin), (d.shape, o.shape)
    label, work = _mm_work("dgrad", hw, cin, cout, dz_idxs is not None, [d.shape[0] for d in dzs], act=acts is not None)
    call("ugn_mm_conv3x3_dgrad_multi", ptr_array([d.data for d in dzs]), ptr_array(dz_idxs) if dz_idxs is not None else None,
         ptr_array([d.meta for d in dzs]), ptr_array(wpks), ptr_array(wmetas),
         ptr_array([a.data for a in acts]) if acts is not None else None, ptr_array([o.data for o in outs]),
         ptr_array([o.meta for o in outs]), ns, len(dzs), hw, cin, cout, _stream(), label=label, work=work)
    return outs


_WS = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(max(int(nbytes), 1), dtype=U8, device=device)
    return buf


def conv3x3_wgrad_mm_multi(xs, dzs, cout, dws, dz_idxs=None):
    """Weight gradients (fp32 HWIO, written to dws) of up to 6 jobs of one shape in one launch.  xs: H2 layer inputs
    [n,hw,hw,cin]; dzs: H2 gradients w.r.t. the pre-activation ([n,hw,hw,cout], or pooled + dz_idxs argmax bytes)."""
    n_, hw, _, cin = xs[0].shape
    ns = (C.c_int * len(xs))(*[x.shape[0] for x in xs])
    nbytes = _lib.load().ugn_mm_conv3x3_wgrad_ws(hw, cin, cout)
    if nbytes == 0:
        raise ValueError("conv3x3_wgrad_mm_multi: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _workspace(nbytes, xs[0].device)
    pooled = dz_idxs is not None
    flops = 2.0 * 9 * cin * cout * hw * hw * sum(x.shape[0] for x in xs)
    # pooled layers run on the SPARSE matrix pipe (v_smfmac_f32_16x16x64_f16: the un-pooled gradient is 2:4 structured along the pixels):
    # half the dense instruction count, so the matrix-pipe time they are priced with is that of 1.5x the algorithmic FLOPs
    kern = "wgrad_mm_kernel<%d, %d, %d, %d, %d>" % (cin, cout, hw, int(pooled), int(pooled))
    pipe = 1.5 if pooled else 3.0
    label = "conv3x3_wgrad[%d->%d @%dx%d%s h2] %s" % (cin, cout, hw, hw, " pooled" if pooled else "", kern)
    call("ugn_mm_conv3x3_wgrad_multi", ptr_array([x.data for x in xs]), ptr_array([x.meta for x in xs]),
         ptr_array([d.data for d in dzs]), ptr_array(dz_idxs) if pooled else None, ptr_array([d.meta for d in dzs]),
         ptr_array(dws), ns, len(xs), hw, cin, cout, ptr(ws), ws.numel(), _stream(), label=label,
         work=dict(flops=flops, mfma_flops=pipe * flops, bytes=_mm_bytes("wgrad", hw, cin, cout, pooled, sum(x.shape[0] for x in xs)),
                   kernel=kern, bound="roof", images=int(sum(x.shape[0] for x in xs)), dtype="f16x2"))
    return dws


# ---- the steps around the 3x3 layers -----------------------------------------------------------------------------------
def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _sizes(v):
    return (C.c_size_t * len(v))(*[int(x) for x in v])


def _opt(ts, attr=None):
    if ts is None:
        return None
    return ptr_array([None if t is None else (getattr(t, attr) if attr else t) for t in ts])


def _hbm(kernel, nbytes):
    return dict(flops=None, mfma_flops=None, bytes=float(nbytes), kernel=kernel, bound="hbm")


def absmax_multi(xs, metas):
    call("ugn_absmax_multi", ptr_array(xs), _sizes([x.numel() for x in xs]), ptr_array(metas), len(xs), _stream(),
         label="absmax", work=_hbm("absmax_multi_kernel", sum(x.numel() for x in xs) * 4.0))
    return metas


def encode_multi(xs, amax_metas, outs):
    """fp32 NHWC tensors -> H2 (outs), given metas that already hold max|x| (absmax_multi)."""
    c = xs[0].shape[-1]
    call("ugn_h2_encode_multi", ptr_array(xs), ptr_array(amax_metas), ptr_array([o.data for o in outs]),
         ptr_array([o.meta for o in outs]), _sizes([x.numel() // c for x in xs]), len(xs), c, _stream(),
         label="h2_encode", work=_hbm("encode_multi_kernel", sum(x.numel() for x in xs) * 8.0))
    return outs


def conv5x5_in_fwd_h2(x, x_meta, w, out, sign=None):
    """First layer with a1 written as H2 (out: H2Tensor [n,64,64,32]); x_meta = {0, bits(max|x|)}."""
    n, cin = x.shape[0], x.shape[3]
    call("ugn_conv5x5_in_fwd_h2", ptr(x), ptr(x_meta), ptr(w), ptr(out.data), ptr(out.meta), ptr(sign), n, cin, _stream(),
         label="conv5x5_fwd[cin=%d h2]" % cin,
         work=_hbm("conv5x5_fwd_kernel<%d, %s, true>" % (cin, "true" if sign is not None else "false"),
                   n * (3600.0 * cin * 4 + 4096 * 32 * 4 + (4096 * 4 if sign is not None else 0))))
    return out


def dgrad32_wgrad5_multi(dzs, dz_idxs, wpks, wmetas, xs, x_metas, signs, dws, scales):
    """Data gradient of the pooled 32 -> 32 layer fused with the 5x5 first layer's weight gradient (csrc/conv3x3_mm.hip
    dgrad32_w5_kernel): dws[j] = dL/dW1 [5,5,cin,32]; dL/da1 is never materialised.  dzs: dL/dp2 (H2, pooled) + argmax bytes; xs:
    network inputs fp32 [n,60,60,cin]; x_metas: {0, bits(max|x|)}; signs: a1 sign words; scales: scratch meta records."""
    njobs = len(dzs)
    ns = [d.shape[0] for d in dzs]
    cins = [x.shape[3] for x in xs]
    nbytes = _lib.load().ugn_mm_dgrad32_wgrad5_ws(njobs)
    ws = _workspace(nbytes, xs[0].device)
    n = int(sum(ns))
    flops = 2.0 * 9 * 32 * 32 * 64 * 64 * n
    label = "conv3x3_dgrad[32->32 @64x64 pooled h2]+conv5x5_wgrad dgrad32_w5_kernel"
    work = dict(flops=flops, mfma_flops=3.0 * flops, bytes=None, kernel="dgrad32_w5_kernel", bound="mfma", images=n, dtype="f16x2")
    call("ugn_mm_dgrad32_wgrad5_multi", ptr_array([d.data for d in dzs]), ptr_array([d.meta for d in dzs]), ptr_array(dz_idxs),
         ptr_array(wpks), ptr_array(wmetas), ptr_array(xs), ptr_array(x_metas), ptr_array(signs), ptr_array(dws), ptr_array(scales),
         _ints(ns), _ints(cins), njobs, ptr(ws), ws.numel(), _stream(), label=label, work=work)
    return dws


# UGN_C5_WGRAD_F16=0: the first layer's weight gradient on the fp32 MFMA (rounds 1-3) instead of the f16 matrix pipe
C5_WGRAD_F16 = __import__("os").environ.get("UGN_C5_WGRAD_F16", "1") != "0"


def conv5x5_in_wgrad_h2(x, dz1, dw, sign=None, x_meta=None):
    """dL/dW of the 5x5 first layer from dz1 = dL/da1 (H2) and the network input x; x_meta = {0, bits(max|x|)} selects the f16-pipe kernel."""
    n, cin = x.shape[0], x.shape[3]
    nbytes = _lib.load().ugn_conv5x5_in_wgrad_ws(n, cin)
    ws = _workspace(nbytes, x.device)
    if x_meta is not None and C5_WGRAD_F16:
        call("ugn_conv5x5_in_wgrad_h2x", ptr(x), ptr(x_meta), ptr(dz1.data), ptr(dz1.meta), ptr(sign), ptr(dw), n, cin, ptr(ws), ws.numel(),
             _stream(), label="conv5x5_wgrad[cin=%d h2]" % cin,
             work=_hbm("conv5x5_wgrad_kernel<%d, %s, 3>" % (cin, "true" if sign is not None else "false"),
                       n * (3600.0 * cin * 4 + 4096 * 32 * 4 + (4096 * 4 if sign is not None else 0))))
        return dw
    call("ugn_conv5x5_in_wgrad_h2", ptr(x), ptr(dz1.data), ptr(dz1.meta), ptr(sign), ptr(dw), n, cin, ptr(ws), ws.numel(), _stream(),
         label="conv5x5_wgrad[cin=%d h2]" % cin,
         work=_hbm("conv5x5_wgrad_kernel<%d, %s, true>" % (cin, "true" if sign is not None else "false"),
                   n * (3600.0 * cin * 4 + 4096 * 32 * 4 + (4096 * 4 if sign is not None else 0))))
    return dw


def setmax_fwd_h2_multi(ps, bs, l, ms=None, addends=None, sums=None, routes=None):
    """H2 set pooling: ms[j] (optional) = max over the l frames, sums[j] = ms[j] + addends[j] (H2 outputs).  routes (optional):
    int32 tensors [b, h, w, 2, c] that receive the routing words (which frames hold the maximum / are positive)."""
    n, h, w, c = ps[0].shape
    extra = 8.0 if routes is not None else 0.0
    call("ugn_h2_setmax_fwd_routed_multi", ptr_array([p.data for p in ps]), ptr_array([p.meta for p in ps]), _opt(addends, "data"),
         _opt(addends, "meta"), _opt(ms, "data"), _opt(ms, "meta"), _opt(sums, "data"), _opt(sums, "meta"), _opt(routes), _ints(bs),
         len(ps), l, h * w, c, _stream(), label="setmax_fwd[%dx%dx%d h2]" % (h, w, c),
         work=_hbm("setmax_fwd_h2_kernel<false>", sum(bs) * ((l + 1.0) * 4 + extra) * h * w * c))
    return ms, sums


def setmax_fwd_h2_f32_multi(ps, bs, l, ms, addends, sums, routes=None):
    """H2 frames (+ H2 set-level addend) -> fp32 maxima ms and sums (the inputs of HPP)."""
    n, h, w, c = ps[0].shape
    extra = 8.0 if routes is not None else 0.0
    call("ugn_h2_setmax_fwd_f32_routed_multi", ptr_array([p.data for p in ps]), ptr_array([p.meta for p in ps]), _opt(addends, "data"),
         _opt(addends, "meta"), _opt(ms), _opt(sums), _opt(routes), _ints(bs), len(ps), l, h * w, c, _stream(),
         label="setmax_fwd[%dx%dx%d h2->f32]" % (h, w, c),
         work=_hbm("setmax_fwd_h2_kernel<true>", sum(bs) * ((l + 1.0) * 4 + extra) * h * w * c))
    return ms, sums


def setmax_bwd_h2_multi(ps, dms, dm_metas, bs, l, lrelu, outs, addends=None, dm_is_f32=False, routes=None):
    """outs[j] (H2Tensor; its data may be addends[j].data) = ((p == max ? dm / #maxima : 0) + addend) * LeakyReLU'(p).
    dms: H2Tensors, or with dm_is_f32 fp32 tensors whose dm_metas hold {0, bits(max|dm|)}.  routes: the routing words of the
    forward pass -- with them the frames ps are not read (same results)."""
    n, h, w, c = outs[0].shape
    dm_ptrs = ptr_array(dms) if dm_is_f32 else ptr_array([d.data for d in dms])
    tail = (int(bool(dm_is_f32)), _opt(addends, "data"), _opt(addends, "meta"), ptr_array([o.data for o in outs]),
            ptr_array([o.meta for o in outs]), _ints(bs), len(outs), l, h * w, c, int(bool(lrelu)), _stream())
    label = "setmax_bwd[%dx%dx%d h2%s]" % (h, w, c, " +addend" if addends is not None else "")
    streams = 2 if addends is not None else 1         # frame-sized tensors moved besides the frames themselves
    if routes is not None:
        call("ugn_h2_setmax_bwd_routed_multi", ptr_array(routes), dm_ptrs, ptr_array(dm_metas), *tail, label=label,
             work=_hbm("setmax_bwd_h2_kernel<%s, true>" % ("true" if dm_is_f32 else "false"),
                       sum(bs) * h * w * c * (l * 4.0 * streams + 12.0)))
    else:
        call("ugn_h2_setmax_bwd_multi", ptr_array([p.data for p in ps]), ptr_array([p.meta for p in ps]), dm_ptrs, ptr_array(dm_metas),
             *tail, label=label, work=_hbm("setmax_bwd_h2_kernel<%s, false>" % ("true" if dm_is_f32 else "false"),
                                           sum(bs) * l * h * w * c * 4.0 * (streams + 1)))
    return outs


def lrelu_bwd_h2_multi(gs, acts, outs):
    n, h, w, c = gs[0].shape
    call("ugn_h2_lrelu_bwd_multi", ptr_array([g.data for g in gs]), ptr_array([g.meta for g in gs]), ptr_array([a.data for a in acts]),
         ptr_array([o.data for o in outs]), ptr_array([o.meta for o in outs]), _sizes([g.shape[0] * h * w for g in gs]), len(gs), c,
         _stream(), label="lrelu_bwd[h2]", work=_hbm("lrelu_bwd_h2_kernel", sum(g.data.numel() for g in gs) * 2.0 * 3))
    return outs


def hpp_bwd_b4h2_multi(as_, s3s, b4s, dfeats, dm3s, dzb4s):
    """HPP backward with b4 as H2Tensors (sign only); dm3s / dzb4s fp32 outputs."""
    bs = [a.shape[0] for a in as_]
    call("ugn_hpp_bwd_b4h2_multi", ptr_array(as_), ptr_array(s3s), ptr_array([b.data for b in b4s]), ptr_array(dfeats), ptr_array(dm3s),
         ptr_array(dzb4s), _ints(bs), len(as_), _stream(), label="hpp_bwd[h2]",
         work=_hbm("hpp_bwd_kernel<true>", sum(bs) * 256 * 128 * 4.0 * 5))
    return dm3s, dzb4s
