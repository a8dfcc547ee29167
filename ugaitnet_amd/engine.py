"""Device-side execution of the UGaitNet gaitset hot path: kernel sequencing and HBM buffer ownership.

This is the host half of the path SURVEY.md section 8(a) rows A1-A16: it owns the parameters (one flat fp32 buffer,
so Adam and the RCCL gradient all-reduce are one launch / one collective per bucket), the saved activations, and
the order in which the HIP kernels of libugaitnet_hip.so run.  torch is used for allocation, streams and
torch.distributed only; every arithmetic op is a C-ABI call (ugaitnet_amd.ops).

Reference call sites mirrored: nets/mj_uwyhNets_ba.py:419-484 (encoder), :668-935 and :1031-1299 (assembly),
nets/triplet_loss_all.py:8-77 (loss), mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:342-349 (data parallel).
"""
from __future__ import annotations

import math

import numpy as np
import torch

import contextlib
import os

from . import _lib, dp, ops, x3
from .engine_bf import BFState, backward_bf, forward_bf
# (the opt-in f16x2 set -- ugaitnet_amd/h2.py, engine_h2.py -- is imported where conv_precision="h2" asks for it: the default library
#  is built without its kernels, python -m ugaitnet_amd.build --h2 adds them)

F32 = torch.float32

# name, kernel, cin (None = modality channels), cout, spatial size, pooled -- creation order of the reference's Conv2D
# layers (nets/mj_uwyhNets_ba.py:428-462), which is also the Keras weight order.
CONV_SPECS = (
    ("a1", 5, None, 32, 64, False), ("a2", 3, 32, 32, 64, True),
    ("b1", 3, 32, 64, 32, False), ("b2", 3, 64, 64, 32, True),
    ("a3", 3, 32, 64, 32, False), ("a4", 3, 64, 64, 32, True),
    ("b3", 3, 64, 128, 16, False), ("b4", 3, 128, 128, 16, False),
    ("a5", 3, 64, 128, 16, False), ("a6", 3, 128, 128, 16, False),
)
NBINS, FEAT, HIDDEN = 62, 128, 256

# Every launch / arithmetic switch of a model lives in ONE Settings object per GaitCore (ugaitnet_amd/config.py); DEFAULTS is what
# the UGN_* environment says at import.  `GaitCore(config=..., conv_precision=...)` takes its own copy: two cores with different
# settings in one process behave like two processes (tests/test_engine_gpu.py::test_two_cores_with_different_settings_in_one_process).
from .config import Settings

DEFAULTS = Settings.from_env()
DEFAULT_PRECISION = DEFAULTS.conv_precision     # (read-only conveniences for callers that only want to know the defaults)
INFER_PRECISION = DEFAULTS.infer_precision
WINO_DGRAD = ("a2", "a3", "a4", "a5", "a6", "b1", "b2", "b3", "b4")


class _Launch:
    """Streams of ONE core and the settings its launches follow: weight gradients on a second stream beside the data gradients (a
    layer's weight gradient and data gradient only share their inputs, so the two persistent launches may overlap -- the tail of one
    and the prologue of the other fill each other's idle CUs; results are unchanged), optional per-branch / forward side streams."""

    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self.wstream, self.bstream = {}, {}

    def wgrad_stream(self):
        # one weight-gradient stream per stream that issues backward chains
        key = torch.cuda.current_stream(self.device).cuda_stream
        st = self.wstream.get(key)
        if st is None:
            st = self.wstream[key] = torch.cuda.Stream(device=self.device)
        return st

    def branch_stream(self, mi):
        st = self.bstream.get(mi)
        if st is None:
            st = self.bstream[mi] = torch.cuda.Stream(device=self.device)
        return st

    @contextlib.contextmanager
    def side(self, device=None):
        """`with launch.side():` -- run the enclosed launches on the weight-gradient stream, after everything issued so far."""
        if not self.cfg.wgrad_stream:
            yield
            return
        st = self.wgrad_stream()
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            yield

    def join_backward_streams(self):
        """The current stream waits for every backward chain and weight gradient issued so far."""
        cur = torch.cuda.current_stream(self.device)
        for st in list(self.bstream.values()) + list(self.wstream.values()):
            cur.wait_stream(st)


@contextlib.contextmanager
def serial_launches(core):
    """Every launch of `core` on ONE stream (no weight-gradient / branch / forward side streams) while the context is active: the
    per-kernel durations bench.py's roofline pass and the rocprofv3 kernel-trace summaries quote.  Results are unchanged."""
    with core.serial_launches():
        yield


def glorot_uniform(gen, shape):
    """keras GlorotUniform (receptive field = prod(shape[:-2]) for rank > 2), host-side numpy."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    lim = math.sqrt(6.0 / ((shape[-2] + shape[-1]) * rf))
    return gen.uniform(-lim, lim, size=shape).astype(np.float32)


def branch_param_shapes(cin):
    shapes = [(name, (k, k, cin if ci is None else ci, co)) for name, k, ci, co, _, _ in CONV_SPECS]
    shapes.append(("fc", (NBINS, FEAT, HIDDEN)))
    return shapes


class ParamStore:
    """All trainable parameters in ONE flat HBM buffer (+ flat grad / Adam m / Adam v), with named views."""

    def __init__(self, named_shapes, device):
        self.device = device
        self.names = [n for n, _ in named_shapes]
        self.shapes = dict(named_shapes)
        self.offsets = {}
        off = 0
        for n, s in named_shapes:
            self.offsets[n] = off
            off += int(np.prod(s))
            off = (off + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = off
        self.before_write = None
        self.flat = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        self.m = torch.zeros(off, dtype=F32, device=device)
        self.v = torch.zeros(off, dtype=F32, device=device)
        self.p = {n: self._view(self.flat, n) for n in self.names}
        self.g = {n: self._view(self.grad, n) for n in self.names}

    def _view(self, buf, n):
        k = int(np.prod(self.shapes[n]))
        return buf[self.offsets[n]:self.offsets[n] + k].view(self.shapes[n])

    def set(self, name, array):
        if self.before_write is not None:      # (GaitCore: a filter repack on the second stream may still be READING the flat buffer)
            self.before_write()
        self.p[name].copy_(torch.as_tensor(np.ascontiguousarray(array), dtype=F32))

    def get(self, name):
        return self.p[name].detach().cpu().numpy()


class Encoder:
    """One modality branch: build_gaitset_branch (nets/mj_uwyhNets_ba.py:419-484), forward and backward."""

    def __init__(self, store, prefix, cin, bf16=False, cfg=None, launch=None):
        self.store, self.prefix, self.cin = store, prefix, cin
        self.cfg = DEFAULTS if cfg is None else cfg              # the owning core's settings (a stand-alone encoder: the defaults)
        self.launch = _Launch(self.cfg, store.device) if launch is None else launch
        # bf16: the 3x3 convolutions, data gradients and weight gradients multiply bf16-rounded Winograd-domain
        # operands (fp32 accumulate, fp32 tensors); the 5x5 layer and everything else stay fp32
        self.bf16 = bool(bf16)
        if self.bf16:
            raise ValueError("the bf16-operand Winograd kernels on fp32 tensors were retired in round 5 (conv_precision='bf16' is configs[4])")
        self.x3 = False    # 3x3 layers on the x3 kernels (GaitCore(conv_precision="f32x3")): fp32 tensors, three-way bf16 split
        self.xf = {}       # x3: packed filter planes, forward
        self.xd = {}       # x3: packed filter planes, data gradient
        self.wp = {}       # packed forward weights of the 3x3 layers (direct kernels)
        self.uf = {}       # Winograd-transformed filters, forward
        self.ud = {}       # Winograd-transformed filters, data gradient
        self.act = None    # saved activations of the last forward
        self.shape = None
        self.scratch = {}

    def _wgrad3x3(self, *a, **kw):
        return (ops.conv3x3_wgrad_wino if self.cfg.use_winograd else ops.conv3x3_wgrad)(*a, **kw)

    def W(self, name):
        return self.store.p[self.prefix + name]

    def G(self, name):
        return self.store.g[self.prefix + name]

    def pack_jobs(self):
        """(w, u_packed, dgrad, pooled) of every 3x3 layer and direction of this branch, for ops.wino_pack_multi."""
        jobs = []
        for name, k, _, _, _, pool in CONV_SPECS:
            if k != 3:
                continue
            w = self.W(name)
            for store, dgrad in ((self.uf, False), (self.ud, True)):
                if name not in store:
                    store[name] = torch.empty((16 * w.shape[2] * w.shape[3],), dtype=F32, device=w.device)
                jobs.append((w, store[name], dgrad, pool))   # a pooled layer's dgrad takes dz at pooled resolution
        return jobs

    def x3_pack_jobs(self):
        """(w, packed planes, dgrad) of every 3x3 layer and direction of this branch, for x3.pack_multi."""
        jobs = []
        for name, k, _, _, _, _ in CONV_SPECS:
            if k != 3:
                continue
            w = self.W(name)
            for store, dgrad in ((self.xf, False), (self.xd, True)):
                if name not in store:
                    store[name] = x3.packed_empty(w.shape[2], w.shape[3], w.device)
                jobs.append((w, store[name], dgrad))
        return jobs

    def repack(self):
        """Refresh the kernel-ready copies of the 3x3 weights (after every optimizer step)."""
        if self.x3:
            x3.pack_multi(self.x3_pack_jobs())
        elif self.cfg.use_winograd:
            ops.wino_pack_multi(self.pack_jobs(), bf16=self.bf16)
        else:
            for name, k, _, _, _, _ in CONV_SPECS:
                if k == 3:
                    self.wp[name] = ops.pack3x3(self.W(name), self.wp.get(name))

    def conv(self, name, x, pool, out, idx=None):
        """3x3 conv + LeakyReLU (+ MaxPool) of layer `name`."""
        if self.cfg.use_winograd:
            return ops.conv3x3_fwd_wino(x, self.uf[name], self.W(name).shape[3], pool, out, idx, bf16=self.bf16)
        return ops.conv3x3_fwd(x, self.wp[name], pool, out, idx)

    def dgrad(self, name, dz, hw, **kw):
        """Data gradient of layer `name` (fused epilogue options as in ops.conv3x3_dgrad)."""
        if self.cfg.use_winograd and name in WINO_DGRAD:
            w = self.W(name)
            return ops.conv3x3_dgrad_wino(dz, self.ud[name], hw, w.shape[2], w.shape[3], bf16=self.bf16, **kw)
        return ops.conv3x3_dgrad(dz, self.W(name), hw, **kw)

    # The global branch applies each 3x3 shape of the frame stack once more, to B set-level maps instead of B*L frames.
    # Alone such a launch fills a fraction of the CUs, so the Winograd kernels take the frame-level layer and its set-level
    # twin (a3|b1, a4|b2, a5|b3, a6|b4) as two jobs of ONE launch wherever the data dependencies allow it.
    def conv_pair(self, names, xs, pool, outs, idxs=None):
        if self.cfg.use_winograd and self.cfg.pair_launches:
            return ops.conv3x3_fwd_wino_pair(list(xs), [self.uf[n] for n in names], self.W(names[0]).shape[3], pool, list(outs),
                                             list(idxs) if pool else None, bf16=self.bf16)
        res = [self.conv(n, x, pool, o, i) for n, x, o, i in zip(names, xs, outs, idxs or (None, None))]
        return ([r[0] for r in res], [r[1] for r in res]) if pool else res

    def dgrad_pair(self, names, dzs, hw, outs, dz_idxs=None, acts=None):
        if self.cfg.use_winograd and self.cfg.pair_launches:
            w = self.W(names[0])
            return ops.conv3x3_dgrad_wino_pair(list(dzs), [self.ud[n] for n in names], hw, w.shape[2], w.shape[3], list(outs),
                                               dz_idxs=dz_idxs, acts=acts, bf16=self.bf16)
        return [self.dgrad(n, dz, hw, dz_idx=None if dz_idxs is None else dz_idxs[k], act=None if acts is None else acts[k],
                           out=outs[k]) for k, (n, dz) in enumerate(zip(names, dzs))]

    def wgrad_pair(self, names, xs, dzs, cout, dz_idxs=None):
        if self.cfg.use_winograd and self.cfg.pair_launches:
            return ops.conv3x3_wgrad_wino_pair(list(xs), list(dzs), cout, [self.G(n) for n in names], dz_idxs=dz_idxs,
                                               bf16=self.bf16)
        return [self._wgrad3x3(x, dz, cout, dz_idx=None if dz_idxs is None else dz_idxs[k], dw=self.G(n))
                for k, (n, x, dz) in enumerate(zip(names, xs, dzs))]

    def _buf(self, pool, key, shape, dtype=F32):
        t = pool.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.store.device)
            pool[key] = t
        return t

    def forward(self, x):
        """x [B,L,60,60,C] (device, fp32) -> [62,B,256]."""
        b, l = x.shape[0], x.shape[1]
        n = b * l
        if self.act is None or self.shape != (b, l):
            self.act, self.shape = {}, (b, l)
        A = self.act
        U8 = torch.uint8
        if not (self.uf if self.cfg.use_winograd else self.wp):
            self.repack()
        xf = x.reshape(n, 60, 60, self.cin)
        A["x"] = xf
        # a1's LeakyReLU' factor travels as one bit per element: the a2 data gradient then skips re-reading a1 (315 MB per
        # modality) and the 5x5 weight gradient applies the factor while it multiplies
        a1 = ops.conv5x5_in_fwd(xf, self.W("a1"), self._buf(A, "a1", (n, 64, 64, 32)),
                                sign=self._buf(A, "a1s", (n, 64, 64), torch.int32) if self.cfg.a1_sign_bits else None)
        p2, i2 = self.conv("a2", a1, True, self._buf(A, "p2", (n, 32, 32, 32)),
                                 self._buf(A, "i2", (n, 32, 32, 32), U8))
        if self.cfg.routed:   # also count the maxima: the set-max gradient is then formed inside the a3 / a5 data-gradient epilogues
            m1, _ = ops.setmax_fwd_cnt(p2, b, l, m=self._buf(A, "m1", (b, 32, 32, 32)), cnt=self._buf(A, "c1", (b, 32, 32, 32)))
        else:
            m1 = ops.setmax_fwd(p2, b, l, m=self._buf(A, "m1", (b, 32, 32, 32)))
        a3, b1 = self.conv_pair(("a3", "b1"), (p2, m1), False,
                                (self._buf(A, "a3", (n, 32, 32, 64)), self._buf(A, "b1", (b, 32, 32, 64))))
        (p4, q2), (i4, j2) = self.conv_pair(("a4", "b2"), (a3, b1), True,
                                            (self._buf(A, "p4", (n, 16, 16, 64)), self._buf(A, "q2", (b, 16, 16, 64))),
                                            (self._buf(A, "i4", (n, 16, 16, 64), U8), self._buf(A, "j2", (b, 16, 16, 64), U8)))
        if self.cfg.routed:
            _, s2, _ = ops.setmax_fwd_cnt(p4, b, l, addend=q2, m=self._buf(A, "m2", (b, 16, 16, 64)),
                                          sum_out=self._buf(A, "s2", (b, 16, 16, 64)), cnt=self._buf(A, "c2", (b, 16, 16, 64)))
        else:
            _, s2 = ops.setmax_fwd(p4, b, l, addend=q2, m=self._buf(A, "m2", (b, 16, 16, 64)),
                                   sum_out=self._buf(A, "s2", (b, 16, 16, 64)))
        a5, b3 = self.conv_pair(("a5", "b3"), (p4, s2), False,
                                (self._buf(A, "a5", (n, 16, 16, 128)), self._buf(A, "b3", (b, 16, 16, 128))))
        a6, b4 = self.conv_pair(("a6", "b4"), (a5, b3), False,
                                (self._buf(A, "a6", (n, 16, 16, 128)), self._buf(A, "b4", (b, 16, 16, 128))))
        m3, s3 = ops.setmax_fwd(a6, b, l, addend=b4, m=self._buf(A, "m3", (b, 16, 16, 128)),
                                sum_out=self._buf(A, "s3", (b, 16, 16, 128)))
        feat = ops.hpp_fwd(m3, s3, self._buf(A, "feat", (NBINS, b, FEAT)))
        return ops.binfc_fwd(feat, self.W("fc"), self._buf(A, "out", (NBINS, b, HIDDEN)))

    def backward(self, dout, scratch):
        """dout [62,B,256] -> parameter gradients written into the store's grad views.
        `scratch` is a dict of frame-sized gradient buffers that may be shared by all branches."""
        A = self.act
        b, l = self.shape
        n = b * l
        S = scratch
        buf = lambda key, shape: self._buf(S, key, shape)
        ops.binfc_bwd(A["feat"], self.W("fc"), dout, self.G("fc"), buf("dfeat", (NBINS, b, FEAT)))
        dm3, dzb4 = ops.hpp_bwd(A["m3"], A["s3"], A["b4"], S["dfeat"], buf("dm3", (b, 16, 16, 128)),
                                buf("dzb4", (b, 16, 16, 128)))
        # block 3 of the frame stack (a5, a6) with block 2 of the global branch (b3, b4)
        dz6 = ops.setmax_bwd(A["a6"], dm3, b, l, True, buf("dz6", (n, 16, 16, 128)))
        dev = self.store.device
        with self.launch.side():
            self.wgrad_pair(("a6", "b4"), (A["a5"], A["b3"]), (dz6, dzb4), 128)
        dz5, dzb3 = self.dgrad_pair(("a6", "b4"), (dz6, dzb4), 16,
                                    (buf("dz5", (n, 16, 16, 128)), buf("dzb3", (b, 16, 16, 128))), acts=(A["a5"], A["b3"]))
        with self.launch.side():
            self.wgrad_pair(("a5", "b3"), (A["p4"], A["s2"]), (dz5, dzb3), 128)
        ds2 = buf("ds2", (b, 16, 16, 64))
        if self.cfg.use_winograd and self.cfg.pair_launches and not self.cfg.routed:
            # Both data gradients of the pair run with a PLAIN epilogue in one launch; what the frame-level one still needs
            # (+ set-max gradient of p4, * LeakyReLU'(p4)) is applied by the set-max backward pass, what the set-level one
            # needs (* LeakyReLU'(q2), the raw copy for the set-max path) by a tiny elementwise kernel.
            raw4, _ = self.dgrad_pair(("a5", "b3"), (dz5, dzb3), 16, (buf("g4", (n, 16, 16, 64)), ds2))
            dq2 = ops.lrelu_bwd(ds2, A["q2"], buf("dq2", (b, 16, 16, 64)))
            dp4 = ops.setmax_bwd(A["p4"], ds2, b, l, True, out=raw4, addend=raw4)
        else:
            dq2 = self.dgrad("b3", dzb3, 16, act=A["q2"], out=buf("dq2", (b, 16, 16, 64)), raw_out=ds2)
            if self.cfg.routed:
                dms2 = ops.div(ds2, A["c2"], buf("dms2", (b, 16, 16, 64)))   # dL/dm2 / #maxima (TF reduce_max gradient)
                dp4 = ops.conv3x3_dgrad_wino_routed(dz5, self.ud["a5"], 16, 64, 128, A["p4"], A["m2"], dms2, l,
                                                    out=buf("g4", (n, 16, 16, 64)))
            elif self.cfg.use_winograd:
                raw4 = self.dgrad("a5", dz5, 16, out=buf("g4", (n, 16, 16, 64)))
                dp4 = ops.setmax_bwd(A["p4"], ds2, b, l, True, out=raw4, addend=raw4)
            else:
                g4 = ops.setmax_bwd(A["p4"], ds2, b, l, False, buf("g4", (n, 16, 16, 64)))
                dp4 = self.dgrad("a5", dz5, 16, act=A["p4"], addend=g4, out=g4)  # in place over the addend
        # block 2 (a3, a4) with block 1 of the global branch (b1, b2); a4 / b2 are pooled: dp4 / dq2 are their gradients at
        # pooled resolution, routed through the argmax maps i4 / j2
        with self.launch.side():
            self.wgrad_pair(("a4", "b2"), (A["a3"], A["b1"]), (dp4, dq2), 64, dz_idxs=(A["i4"], A["j2"]))
        dz3, dzb1 = self.dgrad_pair(("a4", "b2"), (dp4, dq2), 32,
                                    (buf("dz3", (n, 32, 32, 64)), buf("dzb1", (b, 32, 32, 64))),
                                    dz_idxs=(A["i4"], A["j2"]), acts=(A["a3"], A["b1"]))
        with self.launch.side():
            self.wgrad_pair(("a3", "b1"), (A["p2"], A["m1"]), (dz3, dzb1), 64)
        if self.cfg.use_winograd and self.cfg.pair_launches and not self.cfg.routed:
            raw2, dm1 = self.dgrad_pair(("a3", "b1"), (dz3, dzb1), 32, (buf("g2", (n, 32, 32, 32)), buf("dm1", (b, 32, 32, 32))))
            dp2 = ops.setmax_bwd(A["p2"], dm1, b, l, True, out=raw2, addend=raw2)
        else:
            dm1 = self.dgrad("b1", dzb1, 32, out=buf("dm1", (b, 32, 32, 32)))
            if self.cfg.routed:
                dms1 = ops.div(dm1, A["c1"], buf("dms1", (b, 32, 32, 32)))
                dp2 = ops.conv3x3_dgrad_wino_routed(dz3, self.ud["a3"], 32, 32, 64, A["p2"], A["m1"], dms1, l,
                                                    out=buf("g2", (n, 32, 32, 32)))
            elif self.cfg.use_winograd:
                raw2 = self.dgrad("a3", dz3, 32, out=buf("g2", (n, 32, 32, 32)))
                dp2 = ops.setmax_bwd(A["p2"], dm1, b, l, True, out=raw2, addend=raw2)
            else:
                g2 = ops.setmax_bwd(A["p2"], dm1, b, l, False, buf("g2", (n, 32, 32, 32)))
                dp2 = self.dgrad("a3", dz3, 32, act=A["p2"], addend=g2, out=g2)
        # block 1 (a1, a2)
        with self.launch.side():
            self._wgrad3x3(A["a1"], dp2, 32, dz_idx=A["i2"], dw=self.G("a2"), **({"bf16": True} if self.bf16 else {}))
        if self.cfg.a1_sign_bits:
            dz1 = self.dgrad("a2", dp2, 64, dz_idx=A["i2"], out=buf("dz1", (n, 64, 64, 32)))   # dL/da1
            with self.launch.side():
                ops.conv5x5_in_wgrad(A["x"], dz1, self.G("a1"), sign=A["a1s"])
        else:
            dz1 = self.dgrad("a2", dp2, 64, dz_idx=A["i2"], act=A["a1"], out=buf("dz1", (n, 64, 64, 32)))
            with self.launch.side():
                ops.conv5x5_in_wgrad(A["x"], dz1, self.G("a1"))


# One launch per layer for ALL modalities (UGN_MERGE=0: one launch per layer and modality).  The three encoders of
# nets/mj_uwyhNets_ba.py:1102-1140 are the same ten layer shapes, so the Winograd kernels take the frame-level layer and the
# set-level twin of every modality as up to six jobs of one launch: a third of the convolution launches of a step (what a
# 5-clip-per-GPU step of the 8-GPU C4 split mostly consists of), items of all modalities in one work list (no per-modality
# tail), and no need for side streams in the forward pass.

class _Conv3:
    """The 3x3 kernel set the merged forward / backward passes run on: Winograd on the fp32 MFMA (ops.*_wino_multi, transformed
    filters e.uf / e.ud) or x3 (x3.*_multi, packed bf16 planes e.xf / e.xd).  Same tensors, same epilogues, same call shapes."""

    def __init__(self, enc):
        self.x3, self.bf16 = enc.x3, enc.bf16

    def ready(self, e):
        return bool(e.xf) if self.x3 else bool(e.uf)

    def fwd(self, encs, names, xs, cout, pool, outs, idxs=None):
        if self.x3:
            return x3.conv3x3_fwd_multi(xs, [e.xf[n] for n in names for e in encs], cout, pool, outs, idxs)
        return ops.conv3x3_fwd_wino_multi(xs, [e.uf[n] for n in names for e in encs], cout, pool, outs, idxs, bf16=self.bf16)

    def dgrad(self, encs, names, dzs, hw, cin, cout, outs, dz_idxs=None, acts=None):
        if self.x3:
            return x3.conv3x3_dgrad_multi(dzs, [e.xd[n] for n in names for e in encs], hw, cin, cout, outs, dz_idxs=dz_idxs, acts=acts)
        return ops.conv3x3_dgrad_wino_multi(dzs, [e.ud[n] for n in names for e in encs], hw, cin, cout, outs, dz_idxs=dz_idxs, acts=acts,
                                            bf16=self.bf16)

    def wgrad(self, xs, dzs, cout, dws, dz_idxs=None):
        if self.x3:
            return x3.conv3x3_wgrad_multi(xs, dzs, cout, dws, dz_idxs=dz_idxs)
        return ops.conv3x3_wgrad_wino_multi(xs, dzs, cout, dws, dz_idxs=dz_idxs, bf16=self.bf16)


def forward_merged(encs, xs):
    """Encoder.forward of several modality branches in lockstep: [x_m [B_m,L,60,60,C_m]] -> [[62,B_m,256]]."""
    U8 = torch.uint8
    C3, cfg = _Conv3(encs[0]), encs[0].cfg
    geo = []
    for e, x in zip(encs, xs):
        b, l = x.shape[0], x.shape[1]
        if e.act is None or e.shape != (b, l):
            e.act, e.shape = {}, (b, l)
        if not C3.ready(e):
            e.repack()
        geo.append((b, l, b * l))
    A = [e.act for e in encs]
    B = lambda e, A_, key, shape, dtype=F32: e._buf(A_, key, shape, dtype)
    # block 1: the 5x5 layer per modality (its kernel is specialised on the input channels), a2 for all of them
    a1s = []
    for e, x, (b, l, n) in zip(encs, xs, geo):
        xf = x.reshape(n, 60, 60, e.cin)
        e.act["x"] = xf
        a1s.append(ops.conv5x5_in_fwd(xf, e.W("a1"), B(e, e.act, "a1", (n, 64, 64, 32)),
                                      sign=B(e, e.act, "a1s", (n, 64, 64), torch.int32) if cfg.a1_sign_bits else None, x3=C3.x3 and cfg.c5_x3))
    hook = getattr(encs[0], "before_conv3", None)     # (a filter repack queued on the second stream: GaitCore.apply_gradients)
    if hook is not None:
        hook()
    p2s, i2s = C3.fwd(encs, ("a2",), a1s, 32, True, [B(e, e.act, "p2", (g[2], 32, 32, 32)) for e, g in zip(encs, geo)],
                      [B(e, e.act, "i2", (g[2], 32, 32, 32), U8) for e, g in zip(encs, geo)])
    bs, l0 = [g[0] for g in geo], geo[0][1]
    assert all(g[1] == l0 for g in geo), "the modalities of a batch share the set length"
    routed = cfg.set_routed and l0 <= 32
    RW = lambda key, hw, c: [B(e, e.act, key, (g[0], hw * hw * c // 4, 2, 4), torch.int32) for e, g in zip(encs, geo)]

    def setmax(key, ps, ms, hw, c, addends=None, sum_outs=None):
        if routed:
            return ops.setmax_fwd_routed_multi(ps, bs, l0, ms, RW(key, hw, c), addends=addends, sum_outs=sum_outs)
        for e in encs:
            e.act.pop(key, None)
        return ops.setmax_fwd_multi(ps, bs, l0, ms, addends=addends, sum_outs=sum_outs)
    m1s = setmax("r1", p2s, [B(e, e.act, "m1", (g[0], 32, 32, 32)) for e, g in zip(encs, geo)], 32, 32)

    def pair_layer(na, nb, xa, xb, cout, hw, pool, ka, kb, ia=None, ib=None):
        """frame-level layer `na` on xa and set-level twin `nb` on xb, all modalities: jobs = [frame..., set...]"""
        ho = hw // 2 if pool else hw
        outs = [B(e, e.act, ka, (g[2], ho, ho, cout)) for e, g in zip(encs, geo)] + \
               [B(e, e.act, kb, (g[0], ho, ho, cout)) for e, g in zip(encs, geo)]
        idxs = None
        if pool:
            idxs = [B(e, e.act, ia, (g[2], ho, ho, cout), U8) for e, g in zip(encs, geo)] + \
                   [B(e, e.act, ib, (g[0], ho, ho, cout), U8) for e, g in zip(encs, geo)]
        C3.fwd(encs, (na, nb), list(xa) + list(xb), cout, pool, outs, idxs)
        k = len(encs)
        return outs[:k], outs[k:]

    a3s, b1s = pair_layer("a3", "b1", p2s, m1s, 64, 32, False, "a3", "b1")
    p4s, q2s = pair_layer("a4", "b2", a3s, b1s, 64, 32, True, "p4", "q2", "i4", "j2")
    _, s2s = setmax("r2", p4s, [B(e, e.act, "m2", (g[0], 16, 16, 64)) for e, g in zip(encs, geo)], 16, 64, addends=q2s,
                    sum_outs=[B(e, e.act, "s2", (g[0], 16, 16, 64)) for e, g in zip(encs, geo)])
    a5s, b3s = pair_layer("a5", "b3", p4s, s2s, 128, 16, False, "a5", "b3")
    a6s, b4s = pair_layer("a6", "b4", a5s, b3s, 128, 16, False, "a6", "b4")
    m3s, s3s = setmax("r3", a6s, [B(e, e.act, "m3", (g[0], 16, 16, 128)) for e, g in zip(encs, geo)], 16, 128, addends=b4s,
                      sum_outs=[B(e, e.act, "s3", (g[0], 16, 16, 128)) for e, g in zip(encs, geo)])
    feats = ops.hpp_fwd_multi(m3s, s3s, [B(e, e.act, "feat", (NBINS, g[0], FEAT)) for e, g in zip(encs, geo)])
    return ops.binfc_fwd_multi(feats, [e.W("fc") for e in encs], [B(e, e.act, "out", (NBINS, g[0], HIDDEN)) for e, g in zip(encs, geo)])


def backward_merged(encs, douts, scratches):
    """Encoder.backward of several modality branches in lockstep (same arithmetic, one launch per layer for all of them)."""
    C3, cfg, side = _Conv3(encs[0]), encs[0].cfg, encs[0].launch.side
    dev = encs[0].store.device
    geo = [(e.shape[0], e.shape[1], e.shape[0] * e.shape[1]) for e in encs]
    A = [e.act for e in encs]
    k = len(encs)
    buf = lambda i, key, shape: encs[i]._buf(scratches[i], key, shape)
    R = range(k)
    bs, l0 = [g[0] for g in geo], geo[0][1]
    _, dfeats = ops.binfc_bwd_multi([a["feat"] for a in A], [e.W("fc") for e in encs], douts, [e.G("fc") for e in encs],
                                    [buf(i, "dfeat", (NBINS, geo[i][0], FEAT)) for i in R])
    dm3, dzb4 = ops.hpp_bwd_multi([a["m3"] for a in A], [a["s3"] for a in A], [a["b4"] for a in A], dfeats,
                                  [buf(i, "dm3", (geo[i][0], 16, 16, 128)) for i in R], [buf(i, "dzb4", (geo[i][0], 16, 16, 128)) for i in R])
    def setmax_bwd(rkey, pkey, dms, outs, addends=None):
        """reduce_max's gradient (+ addend) * LeakyReLU'(p): from the forward pass's routing words where it wrote them"""
        if all(rkey in a for a in A):
            return ops.setmax_bwd_routed_multi([a[rkey] for a in A], dms, bs, l0, True, outs, addends=addends)
        return ops.setmax_bwd_multi([a[pkey] for a in A], dms, bs, l0, True, outs, addends=addends)
    dz6 = setmax_bwd("r3", "a6", dm3, [buf(i, "dz6", (geo[i][2], 16, 16, 128)) for i in R])

    def wgrad(na, nb, xa, xb, dza, dzb, cout, ia=None, ib=None):
        with side():
            C3.wgrad(list(xa) + list(xb), list(dza) + list(dzb), cout, [e.G(na) for e in encs] + [e.G(nb) for e in encs],
                     dz_idxs=None if ia is None else list(ia) + list(ib))

    def dgrad(na, nb, dza, dzb, hw, cin, cout, outa, outb, ia=None, ib=None, acta=None, actb=None):
        C3.dgrad(encs, (na, nb), list(dza) + list(dzb), hw, cin, cout, list(outa) + list(outb),
                 dz_idxs=None if ia is None else list(ia) + list(ib), acts=None if acta is None else list(acta) + list(actb))
        return outa, outb

    # block 3 of the frame stack (a5, a6) with block 2 of the global branch (b3, b4)
    wgrad("a6", "b4", [a["a5"] for a in A], [a["b3"] for a in A], dz6, dzb4, 128)
    dz5, dzb3 = dgrad("a6", "b4", dz6, dzb4, 16, 128, 128, [buf(i, "dz5", (geo[i][2], 16, 16, 128)) for i in R],
                      [buf(i, "dzb3", (geo[i][0], 16, 16, 128)) for i in R], acta=[a["a5"] for a in A], actb=[a["b3"] for a in A])
    wgrad("a5", "b3", [a["p4"] for a in A], [a["s2"] for a in A], dz5, dzb3, 128)
    # plain epilogue for both data gradients of the pair; the set-max backward pass adds the frame-level extras (+ set-max
    # gradient of p4, * LeakyReLU'(p4)), a small elementwise kernel the set-level one (* LeakyReLU'(q2))
    raw4, ds2 = dgrad("a5", "b3", dz5, dzb3, 16, 64, 128, [buf(i, "g4", (geo[i][2], 16, 16, 64)) for i in R],
                      [buf(i, "ds2", (geo[i][0], 16, 16, 64)) for i in R])
    dq2 = ops.lrelu_bwd_multi(ds2, [a["q2"] for a in A], [buf(i, "dq2", (geo[i][0], 16, 16, 64)) for i in R])
    dp4 = setmax_bwd("r2", "p4", ds2, raw4, addends=raw4)
    # block 2 (a3, a4) with block 1 of the global branch (b1, b2); a4 / b2 are pooled
    i4, j2 = [a["i4"] for a in A], [a["j2"] for a in A]
    wgrad("a4", "b2", [a["a3"] for a in A], [a["b1"] for a in A], dp4, dq2, 64, i4, j2)
    dz3, dzb1 = dgrad("a4", "b2", dp4, dq2, 32, 64, 64, [buf(i, "dz3", (geo[i][2], 32, 32, 64)) for i in R],
                      [buf(i, "dzb1", (geo[i][0], 32, 32, 64)) for i in R], i4, j2, [a["a3"] for a in A], [a["b1"] for a in A])
    wgrad("a3", "b1", [a["p2"] for a in A], [a["m1"] for a in A], dz3, dzb1, 64)
    raw2, dm1 = dgrad("a3", "b1", dz3, dzb1, 32, 32, 64, [buf(i, "g2", (geo[i][2], 32, 32, 32)) for i in R],
                      [buf(i, "dm1", (geo[i][0], 32, 32, 32)) for i in R])
    dp2 = setmax_bwd("r1", "p2", dm1, raw2, addends=raw2)
    # block 1 (a1, a2)
    i2 = [a["i2"] for a in A]
    with side():
        C3.wgrad([a["a1"] for a in A], dp2, 32, [e.G("a2") for e in encs], dz_idxs=i2)
    dz1 = C3.dgrad(encs, ("a2",), dp2, 64, 32, 32, [buf(i, "dz1", (geo[i][2], 64, 64, 32)) for i in R],
                   dz_idxs=i2, acts=None if cfg.a1_sign_bits else [a["a1"] for a in A])
    with side():
        for i, e in enumerate(encs):
            ops.conv5x5_in_wgrad(A[i]["x"], dz1[i], e.G("a1"), sign=A[i]["a1s"] if cfg.a1_sign_bits else None, x3=C3.x3 and cfg.c5_x3)


class GaitCore:
    """Encoders + gate/fusion/normalisation + heads + losses + Adam, for 1, 2 or 3 modalities."""

    def __init__(self, in_channels, nclasses=0, multimodal=None, fuse_mode="sign_max", margin=0.2,
                 loss_weights=(1.0, 1.0), device=None, seed=None, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7,
                 process_group=None, world_size=1, skip_masked=False, dp_mode="replica", conv_precision=None,
                 force_collectives=False, triplet_mode="all", config=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.in_channels = tuple(int(c) for c in in_channels)
        self.nmod = len(self.in_channels)
        self.multimodal = (self.nmod > 1) if multimodal is None else bool(multimodal)
        if not self.multimodal and self.nmod != 1:
            raise ValueError("the single-modality graph takes exactly one input")
        self.nclasses = int(nclasses)
        self.fuse_mode = fuse_mode
        self.margin = float(margin)
        # "all": the compiled loss of the reference, batch-all with its literal flatten -> reshape (nets/triplet_loss_all.py:8-67);
        # "hard": the batch-hard loss `compile_hard` names (nets/mj_uwyhNets_ba.py:1301-1306), per bin
        if triplet_mode not in ("all", "hard"):
            raise ValueError("triplet_mode must be 'all' or 'hard', got %r" % (triplet_mode,))
        self.triplet_mode = triplet_mode
        lw = list(loss_weights) if isinstance(loss_weights, (list, tuple)) else [float(loss_weights)]
        self.loss_weights = (float(lw[0]), float(lw[1]) if len(lw) > 1 else float(lw[0]))
        self.lr, self.beta_1, self.beta_2, self.epsilon = float(lr), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self.pg, self.world = process_group, int(world_size)
        # force_collectives: issue the data-parallel collectives even with one rank (a one-GPU rehearsal of the RCCL calls)
        self.force = bool(force_collectives)
        self.dp_active = self.world > 1 or self.force
        # dp_mode "replica": the reference's MirroredStrategy semantics (normalisation + losses per replica slice, gradients
        # averaged).  "global": the fused features of all replicas are all-gathered, normalisation + losses see the whole
        # batch, gradients are summed -- G replicas x B/G clips then equal one device on B clips (ugaitnet_amd/dp.py).
        if dp_mode not in ("replica", "global"):
            raise ValueError("dp_mode must be 'replica' or 'global', got %r" % (dp_mode,))
        self.dp_mode = dp_mode
        self.global_batch = dp_mode == "global" and self.dp_active
        # skip_masked: run each encoder only on the clips whose modality flag is 1.  A masked (clip, modality) pair is
        # multiplied by 0 in the gate (nets/mj_uwyhNets_ba.py:51-54), so its branch output contributes exactly 0 forward
        # and receives exactly 0 gradient: skipping it changes no result, only the work done.
        self.skip_masked = bool(skip_masked)
        self._active = None
        # frozen_branches: only the classification head trains (what build_or_load(freeze_all=True) leaves trainable on the
        # gaitset path: every layer but the last, `classprob`, gets trainable = False; nets/mj_uwyhNets_ba.py:635-649).  The
        # encoders' backward pass is then not run at all and Adam touches the head's slice of the flat buffer only.
        self.frozen_branches = False

        named = []
        for mi, cin in enumerate(self.in_channels):
            named += [("m%d." % mi + n, s) for n, s in branch_param_shapes(cin)]
        if self.nclasses > 0:
            named += [("head.wc", (NBINS * HIDDEN, self.nclasses)), ("head.bc", (self.nclasses,))]
        self.store = ParamStore(named, self.device)
        # "f32": Winograd fp32 MFMA kernels; "bf16": the same with bf16-rounded MFMA operands; "h2": activations / gradients
        # between the 3x3 layers held as split-fp16 halves + block exponent, 3x3 layers on the f16 matrix pipe at fp32-class
        # accuracy (engine_h2.py, csrc/mm_common.h)
        base = DEFAULTS if config is None else config
        conv_precision, fell_back = base.resolve_precision(conv_precision)
        if fell_back:
            import warnings
            warnings.warn("ugaitnet_amd: UGN_WINO / UGN_PAIR / UGN_MERGE / UGN_A1_BITS / UGN_ROUTED exclude the merged launch path the default "
                          "arithmetic 'f32x3' runs on; no arithmetic was named, so this model uses conv_precision='f32' (the fp32-MFMA sets "
                          "those switches select between).  Name one (UGN_CONV_PRECISION / conv_precision=) to silence this.")
        # this core's OWN copy of the settings (ugaitnet_amd/config.py) and its own streams: nothing process-wide
        self.cfg = base.replace(conv_precision=conv_precision)
        self.launch = _Launch(self.cfg, self.device)
        if conv_precision not in ("f32x3", "f32", "bf16", "h2"):     # ('bf16w', Winograd with bf16-rounded operands on fp32 tensors: retired)
            raise ValueError("conv_precision must be 'f32x3', 'f32', 'h2' or 'bf16', got %r" % (conv_precision,))
        self.conv_precision = conv_precision
        self.h2 = conv_precision == "h2"
        self.bf = conv_precision == "bf16"
        self.x3 = conv_precision == "f32x3"
        if self.x3 and not self.cfg.x3_ok():
            raise ValueError("conv_precision='f32x3' runs on the merged one-launch-per-layer path (UGN_WINO / UGN_PAIR / UGN_MERGE / "
                             "UGN_A1_BITS at their defaults, UGN_ROUTED off); those switches select between the fp32-MFMA kernel sets: "
                             "use them with conv_precision='f32' (UGN_CONV_PRECISION=f32)")
        if (self.h2 or self.bf or self.x3) and self.nmod > 3:
            # one launch per layer carries the frame-level layer and the set-level twin of EVERY modality as jobs, and the kernels'
            # job tables hold 6 (csrc/mm_common.h kMaxJobs; h2_elem.hip / bf_elem.hip kJobs); the reference's graphs stop at 3
            # modalities (nets/mj_uwyhNets_ba.py:1031-1299)
            raise ValueError("conv_precision=%r takes at most 3 modalities (6 jobs per launch), got %d; use conv_precision='f32' with "
                             "UGN_MERGE=0" % (conv_precision, self.nmod))
        self.encoders = [Encoder(self.store, "m%d." % mi, cin, cfg=self.cfg, launch=self.launch) for mi, cin in enumerate(self.in_channels)]
        for enc in self.encoders:
            enc.x3 = self.x3
        if self.bf:
            for enc in self.encoders:
                enc.bf = BFState(enc)
        if self.h2:
            _lib.require_h2("GaitCore(conv_precision='h2')")
        # Settings.persistent_wgs = n (< 256): the persistent launches of EVERY 3x3 set (x3, bf16, f16x2: forward, data gradient, the
        # 5x5 forward; the bf16 / f16x2 weight gradients; the x3 weight gradients keep their 256 fixed shares -- include/ugaitnet_hip.h)
        # leave 256 - n CUs free: room for RCCL's channels when the bucketed all-reduce (ar_overlap) overlaps the backward pass.
        # Results do not depend on it (tests/test_x3_gpu.py).  It is the library's ONE process-wide setting (ugn_set_persistent_wgs):
        # a core that does not ask for a reduced grid sets it back to all CUs, so a later core never inherits an earlier one's grid
        # (round 6: the default x3 arithmetic is covered too -- ADVICE r05).
        if self.h2 or self.bf or self.x3:
            self.persistent_wgs = self.grid_for(self.cfg, self.world)
            ops.set_persistent_wgs(self.persistent_wgs)
        else:
            self.persistent_wgs = 0      # (Winograd fp32-MFMA set: its launches are not persistent)
        if self.h2:
            from . import h2
            from .engine_h2 import H2State
            self.meta_pool = h2.MetaPool(self.device, 64 * self.nmod)
            for enc in self.encoders:
                enc.h2 = H2State(enc, self.meta_pool)
        # the event behind the filter repack that apply_gradients queued on the second stream (None: none outstanding).  Whatever
        # reads the packed filters (the first 3x3 layer), rewrites them (another repack) or writes the parameters they are packed
        # from waits for THIS EVENT on its own current stream -- not for a flag plus a stream looked up by the current stream, which
        # missed a pack queued from another stream and let two packs interleave on the same buffers.
        self._pack_event = None
        self._param_writes = 0              # bumped by every ParamStore.set (a cached inference-arithmetic filter pack goes stale)
        self.store.before_write = self._before_param_write
        for enc in self.encoders:
            enc.before_conv3 = self._before_conv3
        self.scratch = {}
        self.bufs = {}
        self._tri_cache = {}
        # gradient buckets of the flat buffer: one per branch, one for the head
        offs = self.store.offsets
        starts = [offs["m%d.a1" % mi] for mi in range(self.nmod)] + ([offs["head.wc"]] if self.nclasses > 0 else [])
        ends = starts[1:] + [self.store.numel]
        self._buckets = list(zip(starts, ends))
        self._ar_pending = None     # work handles of the bucket all-reduces of the current step
        self.init_weights(seed)

    @staticmethod
    def grid_for(cfg, world):
        """Persistent workgroups per launch for these settings: Settings.persistent_wgs if set; 224 when the bucketed all-reduce runs
        BESIDE the backward pass of more than one rank (RCCL's channels get 32 of the 256 CUs; unmeasured on multi-GPU hardware);
        otherwise 0 = all 256."""
        if cfg.persistent_wgs:
            return int(cfg.persistent_wgs)
        return 224 if (world > 1 and cfg.ar_overlap) else 0

    @contextlib.contextmanager
    def serial_launches(self):
        """Every launch of THIS core on one stream (no weight-gradient / branch / forward side streams) while the context is active: the
        per-kernel durations bench.py's roofline pass and the rocprofv3 kernel-trace summaries quote.  Results are unchanged."""
        saved = (self.cfg.wgrad_stream, self.cfg.branch_streams, self.cfg.fwd_streams)
        self.cfg.wgrad_stream, self.cfg.branch_streams, self.cfg.fwd_streams = False, False, 0
        try:
            yield
        finally:
            self.cfg.wgrad_stream, self.cfg.branch_streams, self.cfg.fwd_streams = saved

    # ---- parameters -------------------------------------------------------------------------------------
    def init_weights(self, seed=None):
        gen = np.random.default_rng(seed)
        for name in self.store.names:
            shape = self.store.shapes[name]
            self.store.set(name, np.zeros(shape, np.float32) if name.endswith(".bc") else glorot_uniform(gen, shape))
        self.weights_changed()

    def _before_param_write(self):
        self._param_writes += 1
        self.join_pack()

    def join_pack(self):
        """The current stream waits for the filter repack queued by the last apply_gradients, if one is outstanding."""
        ev = self._pack_event
        if ev is not None:      # (kept until the NEXT pack replaces it: a consumer on another stream must wait for it too -- ADVICE r04)
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _before_conv3(self):
        """Called by forward_h2 / forward_bf between the 5x5 layer and the first 3x3 layer."""
        self.join_pack()

    def weights_changed(self):
        """Refresh the kernel-ready copies of the 3x3 filters on the current stream and leave the event every consumer waits for."""
        self.join_pack()     # (a repack still running on the second stream: never two packs at once on the pk / wmeta buffers)
        self._repack()
        if torch.cuda.is_current_stream_capturing():
            self._pack_event = None       # (inside a captured graph the pack is ordered by the graph's own edges)
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pack_event = ev

    def _repack(self):
        if self.h2:          # f16 halves + block exponent + L1 bound of every 3x3 filter, both directions: two launches
            from . import h2
            h2.mm_pack_multi([j for e in self.encoders for j in e.h2.pack_jobs()])
            return
        if self.bf:          # bf16 copies of the fp32 master filters in the order the kernels stream them: one launch
            from . import bf16
            bf16.pack_multi([j for e in self.encoders for j in e.bf.pack_jobs()])
            return
        if self.x3:          # the three bf16 planes of every 3x3 filter, both directions: one launch for all branches
            x3.pack_multi([j for e in self.encoders for j in e.x3_pack_jobs()])
            return
        if self.cfg.use_winograd:     # one launch for the filters of all branches
            jobs = [j for e in self.encoders for j in e.pack_jobs()]
            ops.wino_pack_multi(jobs, bf16=self.encoders[0].bf16)
            return
        for e in self.encoders:
            e.repack()

    @contextlib.contextmanager
    def arithmetic(self, precision):
        """`with core.arithmetic("f32"):` -- run the enclosed FORWARD passes with the 3x3 layers in another arithmetic than the model
        trains in; used by the Keras surface for `predict` / `encode` (INFER_PRECISION): a model that trains in f16x2 (one block
        exponent per tensor: what a clip gets depends on the largest clip of its batch, csrc/mm_common.h) answers inference queries
        in IEEE fp32, where an embedding does not depend on the other clips of the batch beyond what the reference's own
        batch-axis normalisation does.  Parameters must not change inside the block (the f16x2 / bf16 filter packs stay valid)."""
        if precision is None or precision == self.conv_precision or self.conv_precision not in ("h2", "bf16"):   # (f32x3: IEEE fp32 tensors already)
            yield                   # (fp32-tensor models answer in their own arithmetic)
            return
        if precision != "f32":
            raise ValueError("arithmetic(%r): only 'f32' inside an 'h2' / 'bf16' model" % (precision,))
        if getattr(self, "_arith_active", False):
            raise RuntimeError("GaitCore.arithmetic() does not nest")
        saved = (self.conv_precision, self.h2, self.bf)
        self.conv_precision, self.h2, self.bf = "f32", False, False
        self._arith_active = True           # forward_backward / apply_gradients raise while the block is active
        try:
            # the Winograd-transformed filters of the current parameters: one launch, skipped while the parameters have not changed
            # since the last block (keyed on the optimizer's step counter and the parameter writes)
            stamp = (self.iterations, self._param_writes)
            if getattr(self, "_arith_pack_stamp", None) != stamp:
                self.weights_changed()
                self._arith_pack_stamp = stamp
            yield
        finally:
            self._arith_active = False
            self.conv_precision, self.h2, self.bf = saved

    def set_params_numpy(self, params):
        """params in the oracle's layout: dict(branches=[{a1..fc}], head={wc,bc})."""
        for mi, bp in enumerate(params["branches"]):
            for k, v in bp.items():
                self.store.set("m%d.%s" % (mi, k), v)
        if "head" in params and self.nclasses > 0:
            self.store.set("head.wc", params["head"]["wc"])
            self.store.set("head.bc", params["head"]["bc"])
        self.weights_changed()

    def get_params_numpy(self):
        out = dict(branches=[{n: self.store.get("m%d.%s" % (mi, n)) for n, _ in branch_param_shapes(c)}
                             for mi, c in enumerate(self.in_channels)])
        if self.nclasses > 0:
            out["head"] = dict(wc=self.store.get("head.wc"), bc=self.store.get("head.bc"))
        return out

    def get_grads_numpy(self):
        g = lambda n: self.store.g[n].detach().cpu().numpy()
        out = dict(branches=[{n: g("m%d.%s" % (mi, n)) for n, _ in branch_param_shapes(c)}
                             for mi, c in enumerate(self.in_channels)])
        if self.nclasses > 0:
            out["head"] = dict(wc=g("head.wc"), bc=g("head.bc"))
        return out

    # ---- helpers ----------------------------------------------------------------------------------------
    def _dev(self, a, shape=None):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        t = t.to(device=self.device, dtype=F32, non_blocking=True).contiguous()
        return t if shape is None else t.reshape(shape)

    def _buf(self, key, shape, dtype=F32):
        t = self.bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs[key] = t
        return t

    def _triplet_lists(self, labels):
        lab = np.asarray(labels).reshape(-1).astype(np.int32)
        key = lab.tobytes()
        hit = self._tri_cache.get(key)
        if hit is None:
            hp, hn, kp, kn = ops.triplet_indices(lab)
            hit = (torch.from_numpy(hp).to(self.device), torch.from_numpy(hn).to(self.device), kp, kn)
            if len(self._tri_cache) > 64:
                self._tri_cache.clear()
            self._tri_cache[key] = hit
        return hit

    def _triplet(self, sig, labels, grad_scale):
        b = sig.shape[1]
        bufs = (self._buf("bin_loss", (NBINS,)), self._buf("bin_num", (NBINS,)), self._buf("dsig", (NBINS, b, HIDDEN)))
        if self.triplet_mode == "hard":
            lab = np.ascontiguousarray(np.asarray(labels).reshape(-1).astype(np.int32))
            key = b"hard" + lab.tobytes()
            dev_lab = self._tri_cache.get(key)
            if dev_lab is None:
                if len(self._tri_cache) > 64:
                    self._tri_cache.clear()
                dev_lab = self._tri_cache[key] = torch.from_numpy(lab).to(self.device)
            return ops.triplet_hard_fwd_bwd(sig, dev_lab, self.margin, grad_scale, *bufs)
        hp, hn, kp, kn = self._triplet_lists(labels)
        return ops.triplet_fwd_bwd(sig, hp, hn, kp, kn, self.margin, grad_scale, *bufs)

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, xs, uses=None, gather=False):
        """xs: list of [B,L,60,60,C_m]; uses: list of [B,1] / [B] (multimodal only).  Returns the signature [62,B,256];
        with gather=True (global-batch data parallelism) B is the batch of all replicas, rank-major."""
        xs = [self._dev(x) for x in xs]
        b = xs[0].shape[0]
        self._active = None
        merged = (self.cfg.merged_ok() and len(self.encoders) > 1) or self.h2 or self.bf or self.x3
        if self.h2:
            self.meta_pool.reset()      # every H2 meta of the step gathers its maximum from zero: one memset
        if self.h2:
            from .engine_h2 import forward_h2 as fwd_many
        else:
            fwd_many = forward_bf if self.bf else forward_merged
        if self.multimodal and self.skip_masked:
            outs, self._active = [None] * self.nmod, []
            sub = []      # (modality, rows tensor or None, input of the active clips)
            for mi, (enc, x) in enumerate(zip(self.encoders, xs)):
                u = uses[mi]
                uh = (u.detach().cpu().numpy() if isinstance(u, torch.Tensor) else np.asarray(u)).reshape(-1)
                rows = np.nonzero(uh != 0)[0]
                if len(rows) == b:
                    self._active.append(None)
                    sub.append((mi, None, x))
                    continue
                idx = torch.from_numpy(rows).to(self.device)
                self._active.append(idx)
                outs[mi] = self._buf("out_full%d" % mi, (NBINS, b, HIDDEN))
                outs[mi].zero_()
                if len(rows):
                    sub.append((mi, idx, ops.gather_rows(x.contiguous(), idx, 0)))
            if merged and (len(sub) > 1 or ((self.h2 or self.bf or self.x3) and sub)):
                res = fwd_many([self.encoders[mi] for mi, _, _ in sub], [x for _, _, x in sub])
            else:
                res = [self.encoders[mi].forward(x) for mi, _, x in sub]
            for (mi, idx, _), o in zip(sub, res):
                if idx is None:
                    outs[mi] = o
                else:
                    ops.scatter_rows(o, idx, 1, outs[mi])
        elif merged:
            outs = fwd_many(self.encoders, xs)
        elif self.cfg.fwd_streams and len(self.encoders) > 1:
            main = torch.cuda.current_stream(self.device)
            outs = [None] * len(self.encoders)
            for mi in range(1, len(self.encoders)):
                st = self.launch.branch_stream(100 + (mi % self.cfg.fwd_streams))
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs[mi] = self.encoders[mi].forward(xs[mi])
            outs[0] = self.encoders[0].forward(xs[0])
            for mi in range(1, len(self.encoders)):
                main.wait_stream(self.launch.branch_stream(100 + (mi % self.cfg.fwd_streams)))
        else:
            outs = [enc.forward(x) for enc, x in zip(self.encoders, xs)]
        self.last_b = b
        self.row0 = dp.group_rank(self.pg) * b if gather else 0      # first row of this replica in the gathered batch
        if not self.multimodal:
            # single-modality graph: no gate, no normalisation (:893-903)
            self.sig = dp.gather_batch_axis(outs[0], 1, self.pg, check=False, force=self.force) if gather else outs[0]
            return self.sig
        self.uses = [self._dev(u, (b,)) for u in uses]
        if not gather and self.cfg.gate_norm_fused and b <= ops.GATE_NORM_MAXB:
            # gate / fMerge and the normalisation over the (local) batch in one launch: same results, one launch and one round trip less
            self.fused, self.sel, self.sig = ops.gate_norm_fwd(outs, self.uses, self.fuse_mode, self._buf("fused", (NBINS, b, HIDDEN)),
                                                               self._buf("sel", (NBINS, b, HIDDEN), torch.uint8),
                                                               self._buf("sig", (NBINS, b, HIDDEN)))
            return self.sig
        self.fused, self.sel = ops.gate_fuse_fwd(outs, self.uses, self.fuse_mode, self._buf("fused", (NBINS, b, HIDDEN)),
                                                 self._buf("sel", (NBINS, b, HIDDEN), torch.uint8))
        if gather:
            self.fused = dp.gather_batch_axis(self.fused, 1, self.pg, check=False, force=self.force)   # (b checked with the labels)
        self.sig = ops.l2norm_batch_fwd(self.fused, self._buf("sig", (NBINS, self.fused.shape[1], HIDDEN)))
        return self.sig

    def _gather_targets(self, labels, onehot):
        """Global-batch mode: labels / one-hot rows of all replicas, rank-major like the gathered features."""
        lab = torch.from_numpy(np.ascontiguousarray(np.asarray(labels).reshape(-1).astype(np.int64))).to(self.device)
        labels = dp.gather_batch_axis(lab, 0, self.pg, force=self.force).cpu().numpy()
        if onehot is not None:
            onehot = dp.gather_batch_axis(self._dev(onehot, (lab.shape[0], -1)), 0, self.pg, force=self.force)
        return labels, onehot

    def predict(self, xs, uses=None):
        """Forward only: (signature [62,B,256], flatten [B,15872], classprob [B,ncls] or None)."""
        sig = self.forward(xs, uses)
        b = sig.shape[1]
        probs = None
        if self.nclasses > 0:
            zeros = self._buf("onehot0", (b, self.nclasses))
            zeros.zero_()
            self.head = ops.head_fwd(sig, self.store.p["head.wc"], self.store.p["head.bc"], zeros, 0.0,
                                     self._head_bufs(b))
            probs = self.head["probs"]
        flat = sig.permute(1, 0, 2).reshape(b, NBINS * HIDDEN)   # transpose [1,0,2] + Flatten (layer 'flatten')
        return sig, flat, probs

    def _head_bufs(self, b):
        n = self.nclasses
        return dict(part=self._buf("part", (4 * NBINS, b, n)), probs=self._buf("probs", (b, n)),
                    row_loss=self._buf("row_loss", (b,)), dlogits=self._buf("dlogits", (b, n)), hit=self._buf("hit", (b,)))

    # ---- training step ----------------------------------------------------------------------------------
    def forward_backward(self, xs, uses, labels, onehot):
        """Forward + loss + full backward; gradients land in store.grad.  Returns device scalars (no sync)."""
        if getattr(self, "_arith_active", False):
            raise RuntimeError("a training step inside GaitCore.arithmetic(): that block runs forward passes in another arithmetic only")
        if self.global_batch:   # (before the forward pass is queued: the label exchange synchronises with the host)
            labels, onehot = self._gather_targets(labels, onehot)
        sig = self.forward(xs, uses, gather=self.global_batch)
        b = sig.shape[1]
        self._ar_pending = [] if (self.dp_active and self.cfg.ar_overlap and not self.cfg.branch_streams) else None
        w_tri, w_id = self.loss_weights
        if self.nclasses > 0:
            # the classification head's forward (two small kernels) runs on the side stream, beside the triplet kernel: both
            # only read the signature; the head's backward ADDS to the triplet's signature gradient, so it waits for both
            oh = self._dev(onehot, (b, self.nclasses))
            with (self.launch.side() if self.cfg.head_side else contextlib.nullcontext()):
                self.head = ops.head_fwd(sig, self.store.p["head.wc"], self.store.p["head.bc"], oh, w_id / b,
                                         self._head_bufs(b))
        self.bin_loss, self.bin_num, dsig = self._triplet(sig, labels, w_tri)
        if self.nclasses > 0:
            if self.cfg.wgrad_stream and self.cfg.head_side:
                torch.cuda.current_stream(self.device).wait_stream(self.launch.wgrad_stream())
            ops.head_bwd(sig, self.store.p["head.wc"], self.head["dlogits"], dsig, True, self.store.g["head.wc"],
                         self.store.g["head.bc"])
            if self.global_batch:   # every replica holds the head gradient of the whole batch; the all-reduce sums them
                ops.scale_(self.store.g["head.wc"], 1.0 / self.world)
                ops.scale_(self.store.g["head.bc"], 1.0 / self.world)
            if not self.frozen_branches:
                self._reduce_bucket(self.nmod)
        if self.frozen_branches:
            if self.nclasses == 0:
                raise ValueError("frozen branches without a classification head: nothing is trainable")
            return
        bl, lo = self.last_b, self.row0
        own = (lambda t: t[:, lo:lo + bl].contiguous()) if self.global_batch else (lambda t: t)
        if self.multimodal and not self.global_batch and self.cfg.gate_norm_fused and b <= ops.GATE_NORM_MAXB:
            douts = ops.gate_norm_bwd(self.fused, sig, dsig, self.sel, self.uses, self.fuse_mode,
                                      [self._buf("dout%d" % m, (NBINS, bl, HIDDEN)) for m in range(self.nmod)])
        elif self.multimodal:
            df = own(ops.l2norm_batch_bwd(self.fused, sig, dsig, self._buf("df", (NBINS, b, HIDDEN))))
            douts = ops.gate_fuse_bwd(df, self.sel, self.uses, self.fuse_mode,
                                      [self._buf("dout%d" % m, (NBINS, bl, HIDDEN)) for m in range(self.nmod)])
        else:
            douts = [own(dsig)]
        if self.h2 or self.bf or self.x3 or (self.cfg.merged_ok() and self.nmod > 1 and not self.cfg.branch_streams):
            encs, ds = [], []
            for mi, (enc, d) in enumerate(zip(self.encoders, douts)):
                idx = self._active[mi] if self._active is not None else None
                if idx is None:
                    encs.append(enc)
                    ds.append(d)
                elif idx.numel() == 0:
                    for name, _ in branch_param_shapes(enc.cin):   # no active clip: this branch's gradient is exactly zero
                        enc.G(name).zero_()
                else:
                    encs.append(enc)
                    ds.append(ops.gather_rows(d, idx, 1))
            if self.h2 or self.bf:
                if encs:
                    if self.h2:
                        from .engine_h2 import backward_h2 as bwd_many
                    else:
                        bwd_many = backward_bf
                    bwd_many(encs, ds, self.launch.side)
            elif len(encs) > 1 or (self.x3 and encs):
                backward_merged(encs, ds, [self.scratch.setdefault(self.encoders.index(e), {}) for e in encs])
            elif encs:
                encs[0].backward(ds[0], self.scratch.setdefault(self.encoders.index(encs[0]), {}))
            for mi in range(self.nmod):
                self._reduce_bucket(mi)
            self.launch.join_backward_streams()
            return
        main = torch.cuda.current_stream(self.device)
        for mi, (enc, d) in enumerate(zip(self.encoders, douts)):
            idx = self._active[mi] if self._active is not None else None
            # (frame-sized gradient scratch per branch: the branches' backward chains overlap)
            scratch = self.scratch.setdefault(mi, {})
            if self.cfg.branch_streams and mi > 0:
                st = self.launch.branch_stream(mi)
                st.wait_stream(main)           # the fusion gradient is ready
                ctx = torch.cuda.stream(st)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                if idx is None:
                    enc.backward(d, scratch)
                elif idx.numel() == 0:
                    for name, _ in branch_param_shapes(enc.cin):   # no active clip: this branch's gradient is exactly zero
                        enc.G(name).zero_()
                else:
                    enc.backward(ops.gather_rows(d, idx, 1), scratch)
            self._reduce_bucket(mi)
        self.launch.join_backward_streams()

    def forward_loss_only(self, xs, uses, labels, onehot):
        """Validation step: forward + both losses/metrics, no parameter gradients."""
        if self.global_batch:
            labels, onehot = self._gather_targets(labels, onehot)
        sig = self.forward(xs, uses, gather=self.global_batch)
        b = sig.shape[1]
        self.bin_loss, self.bin_num, _ = self._triplet(sig, labels, 0.0)
        if self.nclasses > 0:
            oh = self._dev(onehot, (b, self.nclasses))
            self.head = ops.head_fwd(sig, self.store.p["head.wc"], self.store.p["head.bc"], oh, 0.0, self._head_bufs(b))

    def _reduce_bucket(self, k):
        """Queue the all-reduce of gradient bucket k behind everything issued so far (both backward streams)."""
        if self._ar_pending is None:
            return
        lo, hi = self._buckets[k]
        with self.launch.side():    # the weight-gradient stream, ordered after the main stream's work up to here
            w = dp.allreduce_sum_async(self.store.grad[lo:hi], self.pg, force=self.force)
        if w is not None:
            self._ar_pending.append(w)

    def finish_gradient_allreduce(self):
        """Complete the step's gradient reduction; returns the factor Adam applies to the summed gradient."""
        if not self.dp_active:
            return 1.0
        if self.frozen_branches:     # only the head's bucket carries a gradient
            lo, hi = self._buckets[self.nmod]
            scale = dp.allreduce_sum_(self.store.grad[lo:hi], self.pg, force=self.force)
            self._ar_pending = None
            return 1.0 if self.global_batch else scale
        if self._ar_pending is not None:
            pending = self._ar_pending
            dp._timed("allreduce_exposed_wait_ms", lambda: [w.wait() for w in pending])   # (what the buckets did not hide)
            self._ar_pending = None
            scale = 1.0 / self.world
        else:
            scale = dp.allreduce_sum_(self.store.grad, self.pg, force=self.force)
        # global mode: the loss already is the whole batch's, the replicas' gradients add up to its gradient
        return 1.0 if self.global_batch else scale

    def apply_gradients(self):
        """Gradient all-reduce over RCCL (data parallel) + keras Adam, one launch over the flat buffer."""
        if getattr(self, "_arith_active", False):
            raise RuntimeError("an optimizer step inside GaitCore.arithmetic(): that block runs forward passes in another arithmetic only")
        scale = self.finish_gradient_allreduce()
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        st = self.store
        if self.frozen_branches:
            lo, hi = self._buckets[self.nmod]
            ops.adam_step(st.flat[lo:hi], st.grad[lo:hi], st.m[lo:hi], st.v[lo:hi], lr_t, self.beta_1, self.beta_2, self.epsilon, scale)
            return                      # (the convolution filters did not change: no repack)
        ops.adam_step(st.flat, st.grad, st.m, st.v, lr_t, self.beta_1, self.beta_2, self.epsilon, scale)
        if (self.h2 or self.bf or self.x3) and self.cfg.wgrad_stream and self.cfg.pack_on_side_stream and not torch.cuda.is_current_stream_capturing():
            # the repack of the 3x3 filters (3 launches, ~50 us whatever the batch) runs on the second stream, beside the next
            # step's input copies and 5x5 layer; the first 3x3 layer waits for it (`_before_conv3`)
            with self.launch.side():
                self.weights_changed()        # (records the event the first 3x3 layer of the next step waits for)
        else:
            self.weights_changed()

    def train_step(self, xs, uses, labels, onehot):
        self.forward_backward(xs, uses, labels, onehot)
        self.apply_gradients()

    def losses(self):
        """Host copies of the last step's losses/metrics (synchronises)."""
        return self.losses_async().result()

    def losses_async(self):
        """The same without synchronising: the copies to (pinned) host memory are queued on the current stream right behind the
        step -- ahead of anything a later step writes into the same device tensors -- and `.result()` waits for THOSE copies only.
        keras_compat's pipelined `fit` reads step k's losses after step k + 1 has been queued, so the GPU never idles on the host."""
        return _PendingLosses(self)


class _PendingLosses:
    def __init__(self, core):
        self.weights, self.has_head = core.loss_weights, core.nclasses > 0
        src = [core.bin_loss] + ([core.head["row_loss"], core.head["hit"]] if self.has_head else [])
        self.host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in src]
        for h, t in zip(self.host, src):
            h.copy_(t, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(core.device))
        self.value = None

    def result(self):
        if self.value is None:
            self.event.synchronize()
            w_tri, w_id = self.weights
            tri = float(self.host[0].numpy().mean())
            out = dict(triplet=tri, loss=w_tri * tri)
            if self.has_head:
                xent = float(self.host[1].numpy().mean())
                acc = float(self.host[2].numpy().mean())
                out.update(xent=xent, acc=acc, loss=w_tri * tri + w_id * xent)
            self.value, self.host = out, None
        return self.value


class GraphedTrainStep:
    """One training step of a GaitCore as two captured HIP graphs (torch.cuda.CUDAGraph is a hipGraph on ROCm): forward + loss
    + backward, and Adam + filter repack; under data parallelism the gradient all-reduce runs between the two replays.

    A step is ~150 kernel launches; at 24 clips per GPU the GPU needs 9 ms for them and the host 2 ms, but at 4-8 clips (the
    8-GPU split of the reference's 40-clip batch is 5 per GPU) the launches themselves set the pace.  Replaying a graph removes
    the per-launch host cost; the arithmetic, the kernels and the two-stream dependency structure (captured as graph edges) are
    exactly those of `GaitCore.train_step`, so parameters stay bit-identical (tests/test_graph_gpu.py).

    The graph is captured for ONE batch geometry: shapes, the equality structure of the labels (which rows share an identity
    -- what the triplet index lists depend on; the sampler's P x K batches all share it) and dense encoders (no skip_masked).
    `step()` copies the new batch into the captured buffers and replays; a batch of another geometry raises ValueError."""

    def __init__(self, core, xs, uses, labels, onehot):
        if core.skip_masked or core.global_batch or core.cfg.branch_streams:
            raise ValueError("graph capture needs the dense step with per-replica losses")
        if core.cfg.ar_overlap and core.world > 1:
            # the bucket all-reduces would be issued inside the capture and a second reduction would follow in step()
            raise ValueError("graph capture and UGN_AR_OVERLAP=1 exclude each other under data parallelism")
        if _lib.PROFILE is not None:
            raise ValueError("per-kernel event timing cannot be captured")
        self.core = core
        dev = core.device
        self.xs = [core._dev(x).clone() for x in xs]
        self.uses = None if uses is None else [core._dev(u).reshape(-1, 1).clone() for u in uses]
        self.onehot = None if onehot is None else core._dev(onehot).clone()
        self.labels = np.asarray(labels).reshape(-1).copy()
        self.pattern = self._pattern(self.labels)
        self.lr = torch.zeros(1, dtype=F32, device=dev)
        cur = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur)
        fwd_streams, core.cfg.fwd_streams = core.cfg.fwd_streams, 0     # the captured forward pass is one chain (the backward keeps its two streams)
        try:
            with torch.cuda.stream(warm):   # allocations, function attributes, triplet lists: all outside the capture
                for _ in range(2):
                    core.forward_backward(self.xs, self.uses, self.labels, self.onehot)
                    core.finish_gradient_allreduce()
            cur.wait_stream(warm)
            torch.cuda.synchronize(dev)
            self.g_fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fb):
                core.forward_backward(self.xs, self.uses, self.labels, self.onehot)
        finally:
            core.cfg.fwd_streams = fwd_streams
        core._ar_pending = None     # (bucket all-reduces are not captured: the reduction runs between the two graphs)
        scale = (1.0 / core.world) if core.world > 1 else 1.0
        self.g_up = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_up):
            ops.adam_step_dev(core.store.flat, core.store.grad, core.store.m, core.store.v, self.lr, core.beta_1, core.beta_2,
                              core.epsilon, scale)
            core.weights_changed()

    @staticmethod
    def _pattern(labels):
        _, first, inv = np.unique(labels, return_index=True, return_inverse=True)
        return np.argsort(np.argsort(first))[inv]     # identities renumbered in order of first appearance

    def step(self, xs, uses, labels, onehot):
        core = self.core
        labels = np.asarray(labels).reshape(-1)
        if labels.shape != self.labels.shape or not np.array_equal(self._pattern(labels), self.pattern):
            raise ValueError("batch geometry differs from the captured one (labels' equality structure)")
        for dst, src in zip(self.xs, xs):
            dst.copy_(core._dev(src).reshape(dst.shape), non_blocking=True)
        if self.uses is not None:
            for dst, src in zip(self.uses, uses):
                dst.copy_(core._dev(src).reshape(dst.shape), non_blocking=True)
        if self.onehot is not None:
            self.onehot.copy_(core._dev(onehot).reshape(self.onehot.shape), non_blocking=True)
        self.g_fb.replay()
        if core.world > 1:
            dp.allreduce_sum_(core.store.grad, core.pg)
        core.iterations += 1
        t = core.iterations
        self.lr.fill_(core.lr * math.sqrt(1.0 - core.beta_2 ** t) / (1.0 - core.beta_1 ** t))
        self.g_up.replay()
