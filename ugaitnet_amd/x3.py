"""Host wrappers of the "x3" 3x3 kernels: IEEE fp32 tensors in HBM, products on the bf16 matrix pipe through the exact three-way
bf16 split of both operands (csrc/x3_common.h, conv3x3_x3.hip, wgrad3x3_x3.hip).  Same tensors, index maps and epilogues as the
fp32 Winograd wrappers in ops.py; only the filters are consumed packed.  Reference call sites: nets/mj_uwyhNets_ba.py:431-462.
torch holds buffers only."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import call, ptr, ptr_array

I16 = torch.int16
U8 = torch.uint8
F32 = torch.float32
MAX_JOBS = 6
PRODUCTS = 6           # bf16 MFMAs per fp32 product (csrc/x3_common.h kProducts); 9 = all partial products (verification)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _opt(ts):
    return None if ts is None else ptr_array(list(ts))


def _chk(t, dtype=F32):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.is_contiguous(), t.dtype)
    return t


class _WS:
    bufs = {}

    @classmethod
    def get(cls, nbytes, device):
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = cls.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = cls.bufs[key] = torch.empty(max(int(nbytes), 1), dtype=U8, device=device)
        return buf


def _work(kind, hw, cin, cout, pooled, ns, kernel, nbytes, products=PRODUCTS):
    """bench.py's roofline bookkeeping: algorithmic FLOPs (direct-convolution count), the FLOPs the bf16 pipe executes for them
    (six partial products per fp32 product; three dense-equivalent ones on the sparse pipe, which skips the structural zeros of a
    MaxPool gradient) and the algorithmic bytes (every tensor read or written once)."""
    n = int(sum(ns))
    flops = 2.0 * 9 * cin * cout * hw * hw * n
    label = "x3_conv3x3_%s[%d->%d @%dx%d%s] %s" % (kind, cin, cout, hw, hw, " pooled" if pooled else "", kernel)
    return label, dict(flops=flops, mfma_flops=flops * products, bytes=float(nbytes), kernel=kernel, bound="roof", images=n, dtype="bf16x3")


def _waves(kc, nc, pool_epilogue, in_pooled):
    """conv_x3_kernel's last template argument (csrc/conv3x3_x3.hip x3_form): 0 = 8 waves, one workgroup per CU; 1 = 4 waves, two
    workgroups per CU (form 2 -- 8 waves at <= 128 registers, two workgroups per CU -- is built by -DUGN_X3_FORM32=2 only)"""
    if nc > 64 or (nc == 64 and not in_pooled and (pool_epilogue or kc > 64)):
        return 0
    return 1


def split(x):
    """fp32 tensor -> int16 tensor [3, numel] of bf16 bit patterns: the three planes the kernels multiply (tests)."""
    _chk(x)
    planes = torch.empty((3, x.numel()), dtype=I16, device=x.device)
    call("ugn_x3_split", ptr(x), ptr(planes), x.numel(), _stream())
    return planes


def packed_empty(cin, cout, device):
    return torch.empty((27 * cin * cout,), dtype=I16, device=device)      # three bf16 planes of the [3,3,cin,cout] filter


def pack_multi(jobs):
    """jobs: list of (w HWIO fp32 [3,3,cin,cout], packed int16 tensor of 27*cin*cout elements, dgrad flag)."""
    for k in range(0, len(jobs), 64):
        part = jobs[k:k + 64]
        call("ugn_x3_pack_multi", ptr_array([_chk(j[0]) for j in part]), ptr_array([_chk(j[1], I16) for j in part]),
             _ints([j[0].shape[2] for j in part]), _ints([j[0].shape[3] for j in part]), _ints([bool(j[2]) for j in part]),
             len(part), _stream())


def pack(w, dgrad):
    pk = packed_empty(w.shape[2], w.shape[3], w.device)
    pack_multi([(w, pk, dgrad)])
    return pk


def conv3x3_fwd_multi(xs, wpks, cout, pool, outs, idxs=None, products=PRODUCTS):
    """Up to 6 forward convolutions of one shape in a single launch: outs[j] = LeakyReLU(conv(xs[j])) (+ MaxPool, argmax bytes)."""
    assert 1 <= len(xs) <= MAX_JOBS and len(wpks) == len(outs) == len(xs) and (not pool or idxs is not None)
    for t in list(xs) + list(outs):
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    ns = [x.shape[0] for x in xs]
    n = sum(ns)
    ho = hw // 2 if pool else hw
    nbytes = n * (hw * hw * cin * 4 + ho * ho * cout * (5 if pool else 4)) + len(xs) * 54 * cin * cout
    kern = "conv_x3_kernel<%d, %d, %d, %d, 0, %d, %d>" % (cin, cout, hw, 1 if pool else 0, _waves(cin, cout, bool(pool), False), products)
    label, work = _work("fwd", hw, cin, cout, pool, ns, kern, nbytes)
    call("ugn_x3_conv3x3_fwd_multi", ptr_array(xs), ptr_array(wpks), ptr_array(outs), ptr_array(idxs) if pool else None, _ints(ns),
         len(xs), hw, cin, cout, int(bool(pool)), int(products), _stream(), label=label, work=work)
    return (outs, idxs) if pool else outs


def conv3x3_dgrad_multi(dzs, wpks, hw, cin, cout, outs, dz_idxs=None, acts=None, products=PRODUCTS):
    """Up to 6 data gradients of the layer cin -> cout at hw x hw in a single launch; dz_idxs: the gradients are POOLED + argmax
    bytes; acts: outs *= LeakyReLU'(acts)."""
    assert 1 <= len(dzs) <= MAX_JOBS and len(wpks) == len(outs) == len(dzs)
    for t in list(dzs) + list(outs):
        _chk(t)
    ns = [d.shape[0] for d in dzs]
    n = sum(ns)
    pooled = bool(dz_idxs) and dz_idxs[0] is not None
    hz = hw // 2 if pooled else hw
    nbytes = n * (hz * hz * cout * (5 if pooled else 4) + hw * hw * cin * (8 if acts else 4)) + len(dzs) * 54 * cin * cout
    kern = "conv_x3_kernel<%d, %d, %d, %d, %d, %d, %d>" % (cout, cin, hw, 3 if acts else 2, int(pooled), _waves(cout, cin, False, pooled), products)
    label, work = _work("dgrad", hw, cin, cout, pooled, ns, kern, nbytes)
    call("ugn_x3_conv3x3_dgrad_multi", ptr_array(dzs), _opt(dz_idxs) if pooled else None, ptr_array(wpks), _opt(acts), ptr_array(outs),
         _ints(ns), len(dzs), hw, cin, cout, int(products), _stream(), label=label, work=work)
    return outs


def conv3x3_wgrad_multi(xs, dzs, cout, dws, dz_idxs=None, products=PRODUCTS):
    """Up to 6 weight gradients of one shape in a single launch: dws[j] HWIO [3,3,cin,cout] = sum xs[j] (x) dzs[j]."""
    assert 1 <= len(xs) <= MAX_JOBS and len(dzs) == len(dws) == len(xs)
    for t in list(xs) + list(dzs) + list(dws):
        _chk(t)
    hw, cin = xs[0].shape[1], xs[0].shape[3]
    ns = [x.shape[0] for x in xs]
    n = sum(ns)
    pooled = bool(dz_idxs) and dz_idxs[0] is not None
    hz = hw // 2 if pooled else hw
    nbytes_ws = _lib.load().ugn_x3_conv3x3_wgrad_ws(hw, cin, cout)
    if nbytes_ws == 0:
        raise ValueError("x3.conv3x3_wgrad_multi: unsupported shape hw=%d cin=%d cout=%d" % (hw, cin, cout))
    ws = _WS.get(nbytes_ws, xs[0].device)
    nbytes = n * (hw * hw * cin * 4 + hz * hz * cout * (5 if pooled else 4)) + len(xs) * 36 * cin * cout
    # pooled layers: the sparse matrix pipe (v_smfmac_f32_16x16x64_bf16) issues HALF the matrix instructions
    kern = "wgrad_x3s_kernel<%d, %d, %d, %d>" % (cin, cout, hw, products) if pooled else "wgrad_x3_kernel<%d, %d, %d, 0, %d>" % (cin, cout, hw, products)
    label, work = _work("wgrad", hw, cin, cout, pooled, ns, kern, nbytes, products=PRODUCTS // 2 if pooled else PRODUCTS)
    call("ugn_x3_conv3x3_wgrad_multi", ptr_array(xs), ptr_array(dzs), _opt(dz_idxs) if pooled else None, ptr_array(dws), _ints(ns),
         len(xs), hw, cin, cout, ptr(ws), ws.numel(), int(products), _stream(), label=label, work=work)
    return dws
