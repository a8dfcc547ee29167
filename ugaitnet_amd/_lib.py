"""ctypes binding of libugaitnet_hip.so (the C ABI declared in include/ugaitnet_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UGN_LIB (experiments only): another build of the same sources (python -m ugaitnet_amd.build --variant NAME ...)
LIB_PATH = os.environ.get("UGN_LIB") or os.path.join(_HERE, "libugaitnet_hip.so")

FUSE_MODES = {"sign_max": 0, "max": 1, "avg": 2}

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/ugaitnet_hip.h one to one
PROTOTYPES = {
    "ugn_abi_version": (_i, []),
    "ugn_last_error": (C.c_char_p, []),
    "ugn_conv5x5_in_fwd": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "ugn_conv5x5_in_wgrad_ws": (_sz, [_i, _i]),
    "ugn_conv5x5_in_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "ugn_pack3x3": (_i, [_p, _p, _i, _i, _p]),
    "ugn_conv3x3_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_dgrad": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_wgrad_ws": (_sz, [_i, _i, _i, _i]),
    "ugn_conv3x3_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "ugn_wino_pack": (_i, [_p, _p, _i, _i, _i, _p]),
    "ugn_wino_pack_multi": (_i, [C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, _p]),
    "ugn_conv3x3_fwd_wino": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_dgrad_wino": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_fwd_wino_pair": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_conv3x3_fwd_wino_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_dgrad_wino_multi": (_i, [C.POINTER(_p)] * 7 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_wgrad_wino_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p, _sz, _i, _p]),
    "ugn_scale": (_i, [_p, _f, _sz, _p]),
    "ugn_setmax_fwd_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _sz, _p]),
    "ugn_setmax_bwd_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _sz, _i, _p]),
    "ugn_setmax_fwd_routed_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _sz, _p]),
    "ugn_setmax_bwd_routed_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _sz, _i, _p]),
    "ugn_lrelu_bwd_multi": (_i, [C.POINTER(_p)] * 3 + [C.POINTER(_sz), _i, _p]),
    "ugn_hpp_fwd_multi": (_i, [C.POINTER(_p)] * 3 + [C.POINTER(_i), _i, _p]),
    "ugn_hpp_bwd_multi": (_i, [C.POINTER(_p)] * 6 + [C.POINTER(_i), _i, _p]),
    "ugn_binfc_fwd_multi": (_i, [C.POINTER(_p)] * 3 + [C.POINTER(_i), _i, _p]),
    "ugn_binfc_bwd_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _p]),
    "ugn_binfc_bwd_parts_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _p]),
    "ugn_conv3x3_dgrad_wino_routed": (_i, [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
    "ugn_conv3x3_dgrad_wino_pair": (_i, [C.POINTER(_p)] * 7 + [C.POINTER(_i), _i, _i, _i, _p]),
    "ugn_conv3x3_wgrad_wino_pair": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _p, _sz, _p]),
    "ugn_conv3x3_wgrad_wino_ws": (_sz, [_i, _i, _i, _i]),
    "ugn_conv3x3_wgrad_wino": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "ugn_setmax_fwd": (_i, [_p, _p, _p, _p, _i, _i, _sz, _p]),
    "ugn_setmax_fwd_cnt": (_i, [_p, _p, _p, _p, _p, _i, _i, _sz, _p]),
    "ugn_div": (_i, [_p, _p, _p, _sz, _p]),
    "ugn_lrelu_bwd": (_i, [_p, _p, _p, _sz, _p]),
    "ugn_setmax_bwd": (_i, [_p, _p, _p, _p, _i, _i, _sz, _i, _p]),
    "ugn_hpp_fwd": (_i, [_p, _p, _p, _i, _p]),
    "ugn_hpp_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _p]),
    "ugn_binfc_fwd": (_i, [_p, _p, _p, _i, _p]),
    "ugn_binfc_bwd": (_i, [_p, _p, _p, _p, _p, _i, _p]),
    "ugn_gate_fuse_fwd": (_i, [C.POINTER(_p), C.POINTER(_p), _i, _i, _p, _p, _i, _p]),
    "ugn_gate_fuse_bwd": (_i, [_p, _p, C.POINTER(_p), C.POINTER(_p), _i, _i, _i, _p]),
    "ugn_l2norm_batch_fwd": (_i, [_p, _p, _i, _p]),
    "ugn_l2norm_batch_bwd": (_i, [_p, _p, _p, _p, _i, _p]),
    "ugn_gate_norm_fwd": (_i, [C.POINTER(_p), C.POINTER(_p), _i, _i, _p, _p, _p, _i, _p]),
    "ugn_gate_norm_bwd": (_i, [_p, _p, _p, _p, C.POINTER(_p), C.POINTER(_p), _i, _i, _i, _p]),
    "ugn_head_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _p]),
    "ugn_head_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "ugn_triplet_indices_host": (_i, [_p, _i, _p, _p, C.POINTER(_i), C.POINTER(_i)]),
    "ugn_triplet_fwd_bwd": (_i, [_p, _p, _p, _i, _i, _f, _p, _p, _p, _f, _i, _p]),
    "ugn_triplet_hard_fwd_bwd": (_i, [_p, _p, _f, _p, _p, _p, _f, _i, _p]),
    "ugn_assemble_modality": (_i, [_p, _i, _p, _i, _i, _f, _f, _f, _f, _f, _f, _p, _p, _p]),
    "ugn_gather_rows": (_i, [_p, _p, _p, _i, _i, _i, _sz, _p]),
    "ugn_scatter_rows": (_i, [_p, _p, _p, _i, _i, _i, _sz, _p]),
    "ugn_knn_ws": (_sz, [_i, _i]),
    "ugn_knn_predict": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "ugn_adam_step": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _p]),
    "ugn_adam_step_dev": (_i, [_p, _p, _p, _p, _sz, _p, _f, _f, _f, _f, _p]),
    "ugn_set_persistent_wgs": (_i, [_i]),
    "ugn_get_persistent_wgs": (_i, []),
    # bf16 tensors in HBM (configs[4])
    "ugn_bf_pack_multi": (_i, [C.POINTER(_p)] * 2 + [C.POINTER(_i)] * 4 + [_i, _p]),
    "ugn_bf_conv3x3_fwd_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_bf_conv3x3_dgrad_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_bf_conv3x3_wgrad_ws": (_sz, [_i, _i, _i]),
    "ugn_bf_conv3x3_wgrad_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p, _sz, _p]),
    "ugn_conv5x5_in_fwd_bf": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "ugn_conv5x5_in_wgrad_bf": (_i, [_p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "ugn_bf_setmax_fwd_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_bf_setmax_fwd_f32_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_bf_setmax_bwd_multi": (_i, [C.POINTER(_p)] * 2 + [_i] + [C.POINTER(_p)] * 2 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_bf_setmax_fwd_routed_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_bf_setmax_fwd_f32_routed_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_bf_setmax_bwd_routed_multi": (_i, [C.POINTER(_p)] * 2 + [_i] + [C.POINTER(_p)] * 2 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_bf_lrelu_bwd_multi": (_i, [C.POINTER(_p)] * 3 + [C.POINTER(_sz), _i, _i, _p]),
    "ugn_bf_convert_multi": (_i, [C.POINTER(_p)] * 2 + [C.POINTER(_sz), _i, _p]),
    "ugn_hpp_bwd_b4bf_multi": (_i, [C.POINTER(_p)] * 6 + [C.POINTER(_i), _i, _p]),
    # "x3": fp32 tensors, 3x3 products through the exact three-way bf16 split on the bf16 matrix pipe
    "ugn_x3_split": (_i, [_p, _p, _sz, _p]),
    "ugn_x3_pack_multi": (_i, [C.POINTER(_p)] * 2 + [C.POINTER(_i)] * 3 + [_i, _p]),
    "ugn_x3_conv5x5_in_fwd": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "ugn_x3_conv5x5_in_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "ugn_x3_conv3x3_fwd_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _i, _i, _p]),
    "ugn_x3_conv3x3_dgrad_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_x3_conv3x3_wgrad_ws": (_sz, [_i, _i, _i]),
    "ugn_x3_conv3x3_wgrad_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _p, _sz, _i, _p]),
}
# the opt-in f16x2 ("H2") set: include/ugaitnet_hip_h2.h, exported only by a library built with `python -m ugaitnet_amd.build --h2`
PROTOTYPES_H2 = {
    "ugn_absmax": (_i, [_p, _sz, _p, _p]),
    "ugn_h2_encode": (_i, [_p, _p, _p, _sz, _i, _p]),
    "ugn_h2_decode": (_i, [_p, _p, _p, _sz, _i, _p]),
    "ugn_mm_pack_multi": (_i, [C.POINTER(_p)] * 3 + [C.POINTER(_i)] * 3 + [_i, _p]),
    "ugn_mm_conv3x3_fwd_multi": (_i, [C.POINTER(_p)] * 7 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_mm_conv3x3_dgrad_multi": (_i, [C.POINTER(_p)] * 8 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_mm_dgrad32_wgrad5_ws": (_sz, [_i]),
    "ugn_mm_dgrad32_wgrad5_multi": (_i, [C.POINTER(_p)] * 10 + [C.POINTER(_i), C.POINTER(_i), _i, _p, _sz, _p]),
    "ugn_conv5x5_in_fwd_h2": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "ugn_conv5x5_in_wgrad_h2": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "ugn_conv5x5_in_wgrad_h2x": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "ugn_absmax_multi": (_i, [C.POINTER(_p), C.POINTER(_sz), C.POINTER(_p), _i, _p]),
    "ugn_h2_encode_multi": (_i, [C.POINTER(_p)] * 4 + [C.POINTER(_sz), _i, _i, _p]),
    "ugn_h2_setmax_fwd_multi": (_i, [C.POINTER(_p)] * 8 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_h2_setmax_fwd_f32_multi": (_i, [C.POINTER(_p)] * 6 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_h2_setmax_bwd_multi": (_i, [C.POINTER(_p)] * 4 + [_i] + [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_h2_setmax_fwd_routed_multi": (_i, [C.POINTER(_p)] * 9 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_h2_setmax_fwd_f32_routed_multi": (_i, [C.POINTER(_p)] * 7 + [C.POINTER(_i), _i, _i, _i, _i, _p]),
    "ugn_h2_setmax_bwd_routed_multi": (_i, [C.POINTER(_p)] * 3 + [_i] + [C.POINTER(_p)] * 4 + [C.POINTER(_i), _i, _i, _i, _i, _i, _p]),
    "ugn_h2_lrelu_bwd_multi": (_i, [C.POINTER(_p)] * 5 + [C.POINTER(_sz), _i, _i, _p]),
    "ugn_hpp_bwd_b4h2_multi": (_i, [C.POINTER(_p)] * 6 + [C.POINTER(_i), _i, _p]),
    "ugn_mm_conv3x3_wgrad_ws": (_sz, [_i, _i, _i]),
    "ugn_mm_conv3x3_wgrad_multi": (_i, [C.POINTER(_p)] * 6 + [C.POINTER(_i), _i, _i, _i, _i, _p, _sz, _p]),
}

_lib = None


class UgnError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the copy already in the process when ours is resolved, or the library binds
    # to a second HIP runtime that knows no device ("no ROCm-capable device is detected" at the first launch)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise UgnError(
            "libugaitnet_hip.so is missing (%s). Build it with `python -m ugaitnet_amd.build`; "
            "ugaitnet_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ugn_abi_version() != 1:
        raise UgnError("libugaitnet_hip.so ABI version mismatch")
    global HAS_H2
    HAS_H2 = hasattr(lib, "ugn_mm_conv3x3_fwd_multi")
    if HAS_H2:
        for name, (res, args) in PROTOTYPES_H2.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


HAS_H2 = None     # set by load(): the library carries the opt-in f16x2 set


def has_h2():
    load()
    return bool(HAS_H2)


def require_h2(what="conv_precision='h2'"):
    if not has_h2():
        raise UgnError("%s needs the opt-in f16x2 kernel set, which this libugaitnet_hip.so was built without: "
                       "`python -m ugaitnet_amd.build --h2` (or UGN_BUILD_H2=1) adds it" % what)


def check(rc, what):
    if rc != 0:
        msg = load().ugn_last_error().decode("utf-8", "replace")
        if rc == -22:
            raise ValueError("%s: %s" % (what, msg))
        raise UgnError("%s failed (code %d): %s" % (what, rc, msg))


# Per-launch timing for bench.py's serialised roofline pass: while PROFILE is a dict, every C-ABI launch is bracketed by a
# HIP-event pair on the stream it is launched on (torch's current stream) and filed under its label.  WORK holds, per
# label, what one such launch does: dict(flops=algorithmic FLOPs, mfma_flops=FLOPs executed on the matrix pipe,
# bytes=algorithmic HBM bytes, kernel=the device kernel rocprofv3 shows for it).  None (the default) = no events at all.
PROFILE = None
WORK = {}
ORDER = None       # a list while the order of the profiled launches is wanted too (bench.py records what it drops with it)


def call(name, *args, label=None, work=None):
    fn = getattr(load(), name)
    if PROFILE is None:
        check(fn(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    key = label or name
    PROFILE.setdefault(key, []).append((e0, e1))
    if ORDER is not None:
        ORDER.append(key)
    if work is not None:
        WORK[key] = work
    check(rc, name)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """Host array of device pointers (None entries -> NULL)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
