"""Counterpart of the reference's nets/triplet_loss_all.py:8-77 for host-side use (validation utilities, tests).

`triplet_loss(margin)(y_true, y_pred)` keeps the reference's closure signature.  y_pred is a [62, B, 256] array
(numpy or a CUDA tensor); the loss is computed by the ugn_triplet_fwd_bwd HIP kernel.  During training the model
object calls the same kernel directly (engine.GaitCore.forward_backward)."""
from __future__ import annotations

import numpy as np


def triplet_loss(margin=1.0):
    def loss(y_true, y_pred):
        import torch
        from .. import ops
        labels = np.asarray(y_true.cpu() if hasattr(y_true, "cpu") else y_true).reshape(-1)
        sig = y_pred if isinstance(y_pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_pred, dtype=np.float32))
        sig = sig.to("cuda", torch.float32).contiguous()
        hp, hn, kp, kn = ops.triplet_indices(labels)
        bl, _, _ = ops.triplet_fwd_bwd(sig, torch.from_numpy(hp).cuda(), torch.from_numpy(hn).cuda(), kp, kn, margin, 0.0)
        return float(bl.cpu().numpy().mean())
    loss.margin = margin
    return loss
