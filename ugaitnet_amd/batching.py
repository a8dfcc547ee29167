"""Device-side batch assembly.

The reference's generator (`data/mj_dataGeneratorMMUWYHsingle_repetitions.py`) decodes every sample on the host to float,
re-lays it out for the gaitset network and repeats it `expand` times with modalities disabled, all in float64 numpy, and
ships the result to the GPU.  Here the host only decides WHICH rows exist (`plan_rows`, the same rules and the same
`random` call sequence as `__gen_batchMM` :776-806); the raw int16 / uint8 sample arrays are uploaded once and
`DeviceBatchAssembler` produces the fp32 `[rows,25,60,60,C]` tensors and `[rows,1]` flags in HBM (ugn_assemble_modality).
"""
from __future__ import annotations

import ctypes as C
import random as _random

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

NOISE = 0.000000001   # reference `self.noise` (:102)


class ModalitySpec:
    """How one modality's `data` array decodes (`__load_dd` :300-318)."""

    def __init__(self, kind, channels, compress_factor=1.0, ntype=2):
        if kind not in ("of", "gray", "depth", "silhouette"):
            raise ValueError("kind must be of | gray | depth | silhouette")
        self.kind, self.channels = kind, int(channels)
        self.is_int16 = kind == "of"
        if kind == "of":      # compressFactor > 1 branch: x / compressFactor (* 0.1 for ntype 2)
            self.divisor, self.offset, self.post_mul = float(compress_factor), 0.0, (0.1 if ntype == 2 else 1.0)
        elif kind == "silhouette":
            self.divisor, self.offset, self.post_mul = 255.0, 0.0, 1.0
        else:
            self.divisor, self.offset, self.post_mul = 255.0, 0.5, 1.0


def plan_rows(present, expand, rng=_random):
    """Row plan of `__gen_batchMM`: present [nbase][nmods] bool (sample file exists) -> src [nbase*expand][nmods] int32,
    the base sample a row's modality copies or -1 (noise, flag 0).  Row i*expand carries every present modality (:732-737);
    the extra rows follow :776-806 -- even i: `ndisable` random modalities off (drawn with replacement), odd i: a single
    modality (i+ex)%3 on.  `rng` must offer randrange like the `random` module the reference uses."""
    present = np.asarray(present, bool)
    nbase, nmods = present.shape
    expand = max(1, int(expand))
    src = np.full((nbase * expand, nmods), -1, np.int32)
    for i in range(nbase):
        for j in range(nmods):
            if present[i, j]:
                src[i * expand, j] = i
        for ex in range(expand - 1):
            if i % 2 == 0:
                ndisable = min(ex + 1, nmods - 1) if expand > 2 else rng.randrange(1, nmods, 1)
                l_dis = [1] * nmods
                for _ in range(ndisable):
                    l_dis[rng.randrange(0, nmods, 1)] = 0
            else:
                l_dis = [0] * nmods
                l_dis[(i + ex) % 3] = 1
            for j in range(nmods):
                # a "copy" of an absent modality copies its noise row: still noise, but the reference sets the flag to 1
                # (:805-806 copy x and write 1.0) -- reproduced by pointing at the base row and letting the caller's
                # `present` decide the payload
                src[(ex + 1) + i * expand, j] = i if l_dis[j] else -1
    return src


def plan_rows_2mod(present, expand, rng=_random):
    """Row plan of the TWO-modality generator `__gen_batch` (:485-528): row i*expand carries what exists; the next row turns one
    randomly drawn modality off and copies the other, a third row (expand > 2) swaps the two; the reference fills no more
    than three rows per sample."""
    present = np.asarray(present, bool)
    nbase, nmods = present.shape
    if nmods != 2:
        raise ValueError("plan_rows_2mod is the two-modality rule")
    expand = max(1, int(expand))
    if expand > 3:
        raise ValueError("the two-modality generator defines at most 3 rows per sample (expand_level <= 3)")
    src = np.full((nbase * expand, 2), -1, np.int32)
    for i in range(nbase):
        for j in range(2):
            if present[i, j]:
                src[i * expand, j] = i
        if expand > 1:
            choice = rng.randrange(0, 2, 1)
            for row in range(1, expand):
                src[row + i * expand, 1 - choice] = i
                choice = 1 - choice
    return src


class DeviceBatchAssembler:
    def __init__(self, specs, device=None):
        self.specs = list(specs)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def assemble(self, raws, src, present=None, clip=(0.0, 0.0)):
        """raws[m]: [nbase,60,60,25*C] int16/uint8 (numpy or device tensor); src: plan_rows output.
        present [nbase][nmods]: a copied-but-absent modality keeps flag 1 with a noise payload, exactly as the reference does.
        clip = (clip_max, clip_min) of the optical-flow augmentation (:723-728), applied to int16 modalities only.
        Returns [(x_m [rows,25,60,60,C] fp32, u_m [rows,1] fp32)] on the device."""
        src = np.asarray(src, np.int32)
        rows = src.shape[0]
        out = []
        for m, spec in enumerate(self.specs):
            raw = raws[m]
            if not isinstance(raw, torch.Tensor):
                raw = torch.from_numpy(np.ascontiguousarray(raw))
            raw = raw.to(self.device).contiguous()
            want = torch.int16 if spec.is_int16 else torch.uint8
            if raw.dtype != want or tuple(raw.shape[1:]) != (60, 60, 25 * spec.channels):
                raise ValueError("modality %d: expected %s [n,60,60,%d], got %s %s" % (m, want, 25 * spec.channels, raw.dtype,
                                                                                       tuple(raw.shape)))
            col = src[:, m].copy()
            flag_fix = None
            if present is not None:
                # An extra row that "copies" a modality whose sample file is absent copies the base row's noise, and the
                # reference still writes flag 1.0 for it (:804-806): noise payload, flag 1.
                pres = np.asarray(present, bool)
                expand = rows // pres.shape[0]
                absent = (col >= 0) & ~pres[np.maximum(col, 0), m]
                flag_fix = absent & ((np.arange(rows) % expand) != 0)
                col[absent] = -1
            if col.max(initial=-1) >= raw.shape[0]:
                raise ValueError("plan refers to base sample %d but only %d were uploaded" % (col.max(), raw.shape[0]))
            scol = torch.from_numpy(col).to(self.device)
            x = torch.empty((rows, 25, 60, 60, spec.channels), dtype=torch.float32, device=self.device)
            u = torch.empty((rows, 1), dtype=torch.float32, device=self.device)
            cmax, cmin = (clip if spec.is_int16 else (0.0, 0.0))
            call("ugn_assemble_modality", ptr(raw), int(spec.is_int16), ptr(scol), rows, spec.channels, spec.divisor, spec.offset,
                 spec.post_mul, float(cmax), float(cmin), NOISE, ptr(x), ptr(u), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if flag_fix is not None and flag_fix.any():
                u[torch.from_numpy(np.nonzero(flag_fix)[0]).to(self.device)] = 1.0
            out.append((x, u))
        return out
