"""Launch / arithmetic settings of a GaitCore, as ONE object per core instead of import-time module globals.

`Settings.from_env()` reads the UGN_* environment switches once (ugaitnet_amd.engine.DEFAULTS); `GaitCore(config=..., **overrides)`
takes its own copy, so two cores with different settings in one process behave like two processes, and `core.serial_launches()`
serialises THAT core's launches only.  Every switch changes scheduling or the kernel set, never a result beyond rounding (each is
covered by a test that says which).
"""
from __future__ import annotations

import dataclasses
import os


@dataclasses.dataclass
class Settings:
    # arithmetic of the 3x3 layers when a model does not name one (UGN_CONV_PRECISION):
    #   "f32x3" (default) IEEE fp32 tensors everywhere; the 3x3 layers multiply them on the bf16 matrix pipe through the exact three-way
    #           bf16 split of both operands (six partial products per fp32 product, fp32 accumulate: csrc/x3_common.h)
    #   "f32"   IEEE fp32 tensors, Winograd F(2x2,3x3) on the fp32 MFMA
    #   "h2"    activations / gradients between the 3x3 layers as split-fp16 halves + ONE block exponent per tensor (22 significant
    #           bits), 3x3 layers on the f16 matrix pipe: fastest, narrower than the reference's fp32, batch-dependent -- opt-in
    #   "bf16"  BASELINE configs[4]: bf16 tensors in HBM, bf16 matrix pipe, fp32 accumulate / master weights / Adam
    conv_precision: str = "f32x3"
    # arithmetic of forward-only queries behind the Keras surface (predict / encode): "f32" or "same" (UGN_INFER_PRECISION)
    infer_precision: str = "f32"
    use_winograd: bool = True        # UGN_WINO=0: the direct fp32-MFMA kernels (first-max on every MaxPool tie) instead of Winograd
    pair_launches: bool = True       # UGN_PAIR: frame-level layer + set-level twin in one launch
    a1_sign_bits: bool = True        # UGN_A1_BITS: LeakyReLU' of the first layer from one bit per element
    wgrad_stream: bool = True        # UGN_WSTREAM: weight gradients on a second stream beside the data gradients
    branch_streams: bool = False     # UGN_BSTREAMS=1: the modalities' backward chains on streams of their own (measured slower)
    fwd_streams: int = 2             # UGN_FSTREAMS: side streams of the un-merged forward pass
    pack_on_side_stream: bool = True  # UGN_PACK_SIDE: the filter repack beside the next step's 5x5 layer
    merge_modalities: bool = True    # UGN_MERGE: one launch per layer for ALL modalities
    ar_overlap: bool = False         # UGN_AR_OVERLAP=1: the gradient all-reduce in four buckets beside the backward pass
    head_side: bool = True           # UGN_HEAD_SIDE: the classification head's forward beside the triplet kernel
    routed: bool = False             # UGN_ROUTED=1: the set-max gradient inside the a3 / a5 data-gradient epilogues (measured slower)
    gate_norm_fused: bool = True     # UGN_GATE_NORM_FUSED: gate / fMerge + batch-axis normalisation in one launch each way
    set_routed: bool = True          # UGN_SET_ROUTED: set-pooling gradients from the forward pass's routing words
    fuse_w5: bool = False            # UGN_FUSE_W5=1 (h2): a2 data gradient fused with the 5x5 weight gradient (measured slower)
    c5_x3: bool = True               # UGN_C5_X3=0: the first layer of the x3 set on the fp32 MFMA instead of the x3 arithmetic (A/B)
    persistent_wgs: int = 0          # UGN_PERSISTENT_WGS: CUs the persistent launches occupy (0 = all; the library's one global)
    precision_named: bool = False    # the arithmetic was NAMED (UGN_CONV_PRECISION set, or GaitCore(conv_precision=...)): never replaced

    @classmethod
    def from_env(cls, env=None):
        e = os.environ if env is None else env
        on = lambda k, d="1": e.get(k, d) != "0"
        s = cls(conv_precision=e.get("UGN_CONV_PRECISION", "f32x3"), infer_precision=e.get("UGN_INFER_PRECISION", "f32"),
                use_winograd=on("UGN_WINO"), pair_launches=on("UGN_PAIR"), a1_sign_bits=on("UGN_A1_BITS"), wgrad_stream=on("UGN_WSTREAM"),
                branch_streams=e.get("UGN_BSTREAMS", "0") == "1", fwd_streams=int(e.get("UGN_FSTREAMS", "2")),
                pack_on_side_stream=on("UGN_PACK_SIDE"), merge_modalities=on("UGN_MERGE"), ar_overlap=e.get("UGN_AR_OVERLAP", "0") == "1",
                head_side=e.get("UGN_HEAD_SIDE", "1") == "1", routed=e.get("UGN_ROUTED", "0") == "1", gate_norm_fused=on("UGN_GATE_NORM_FUSED"),
                set_routed=on("UGN_SET_ROUTED"), fuse_w5=on("UGN_FUSE_W5", "0"), c5_x3=on("UGN_C5_X3"), persistent_wgs=int(e.get("UGN_PERSISTENT_WGS", "0") or 0),
                precision_named="UGN_CONV_PRECISION" in e)
        return s.normalised()

    def normalised(self):
        self.branch_streams = bool(self.branch_streams and self.wgrad_stream)
        self.routed = bool(self.routed and self.use_winograd)
        return self

    def replace(self, **kw):
        return dataclasses.replace(self, **kw).normalised()

    def x3_ok(self):
        """the default arithmetic ("f32x3") runs on the merged one-launch-per-layer path with the first layer's sign bits"""
        return self.merged_ok() and self.a1_sign_bits

    def resolve_precision(self, named=None):
        """The arithmetic a core gets: `named` (GaitCore(conv_precision=...)) or this object's.  The x3 set needs the merged path; when the
        launch switches exclude it (UGN_WINO=0, UGN_PAIR=0, UGN_MERGE=0, UGN_A1_BITS=0, UGN_ROUTED=1 -- the A/B switches of the fp32
        kernel sets) and NOBODY named an arithmetic, the core falls back to "f32", the set those switches belong to, with a warning; a
        NAMED "f32x3" with such switches is an error (`UGN_WINO=0 python mains/...` used to die at construction)."""
        if named is not None:
            return named, False
        if self.conv_precision == "f32x3" and not self.precision_named and not self.x3_ok():
            return "f32", True
        return self.conv_precision, False

    def merged_ok(self):
        """one launch per layer for all modalities is available on the default Winograd pair-launch path"""
        return self.merge_modalities and self.use_winograd and self.pair_launches and not self.routed
