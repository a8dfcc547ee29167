"""`DataGeneratorGaitMMUWYH` with the reference's constructor (data/mj_dataGeneratorMMUWYHsingle_repetitions.py:41-110), backed
by ugaitnet_amd.sampler.DeviceDataGenerator: the mains' `DataGeneratorGaitMMUWYH(allSamples, ..., datadir=[...], labmap=...,
gait=..., nmods=3, gaitset=True, repetition=r, expand_level=e)` keeps working and yields batches that already live in HBM.

Implemented: the one-, two- and three-modality gaitset generators without augmentation (`nmods=1|2|3, gaitset=True,
augmentation_x=0`), the label cycling of `__getitem__`, `__len__`, `on_epoch_end`, `keep_data`.  Everything else the
reference's class can do (affine / mirror augmentation, sample weights, auxiliary / per-FC label lists, 3-D inputs, debug
batches) raises NotImplementedError instead of silently producing different batches."""
from __future__ import annotations

import os

import numpy as np

from ..batching import ModalitySpec
from ..sampler import DeviceDataGenerator
from .. import samples as _samples


class DataGeneratorGaitMMUWYH(DeviceDataGenerator):
    def __init__(self, allSamples, targets=[], batch_size=32, dim=[(50, 60, 60), (25, 60, 60)], n_classes=150, shuffle=True,
                 augmentation=True, datadir=[], sess=None, labmap=[], gait=[], ntype=1, isTest=False, augmentation_x=1,
                 expand_level=2, balanced_classes=True, isTriplet=False, use3D=False, isDebug=False, softlabel=False, camera=[],
                 nmods=2, use_weights=False, meanSample=0.0, aux_losses=False, triplet_all_fc=False, nfcs=0, keep_data=False,
                 gaitset=False, repetition=4):
        if nmods == 1:      # `dim` is one (frames, 60, 60) tuple there (:66-69), the file is allSamples[i][0][0] under datadir[0]
            dim = [tuple(dim)] if not isinstance(dim[0], (tuple, list)) else [tuple(dim[0])]
            datadir = list(datadir[:1])
        if augmentation_x > 0:
            # the reference's default is augmentation_x=1 and its gaitset mains leave it there (mains/mj_trainUWYHGaitNet_
            # DataGen_CasiaB.py:465): name the argument, the random affine augmentation (:706-728) is not built on this path
            raise NotImplementedError("DataGeneratorGaitMMUWYH on the MI355X path: augmentation_x=%r (random affine / mirror "
                                      "augmentation, the reference's default) is not implemented; pass augmentation_x=0"
                                      % (augmentation_x,))
        unsupported = [name for name, bad in (("nmods > 3", nmods not in (1, 2, 3)), ("gaitset=False", not gaitset), ("use3D", use3D),
                                              ("isDebug", isDebug), ("softlabel", softlabel), ("use_weights", use_weights),
                                              ("aux_losses", aux_losses), ("triplet_all_fc", triplet_all_fc),
                                              ("augmentation_x > 0", augmentation_x > 0), ("ntype != 2", ntype != 2)) if bad]
        if unsupported:
            raise NotImplementedError("DataGeneratorGaitMMUWYH on the MI355X path: %s not implemented" % ", ".join(unsupported))
        if len(datadir) != nmods or len(dim) != nmods:
            raise ValueError("one datadir and one dim per modality")
        specs = []
        for m in range(nmods):
            # two spellings of `dim` reach this class: the legacy (frames, 60, 60) with 50 = 25 frames x (x, y) flow, and the
            # gaitset input shape the CASIA-B main passes, (25, 60, 60, channels) with channels = 2 for optical flow
            # (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:213,465: input_shape = [(25, 60, 60, 2), (25, 60, 60, 1)])
            d = tuple(dim[m])
            if len(d) == 4:
                if d[0] != 25 or d[3] not in (1, 2):
                    raise ValueError("dim[%d] must be (25, 60, 60, 1 | 2), got %r" % (m, d))
                frames = 50 if d[3] == 2 else 25
            else:
                frames = d[0]
            if frames not in (25, 50):
                raise ValueError("dim[%d][0] must be 25 or 50 frames, got %r" % (m, frames))
            if frames == 50:      # x/y optical flow interleaved; compressFactor travels in the sample files (:300-309)
                specs.append(ModalitySpec("of", 2, compress_factor=self._compress_factor(allSamples, datadir, m), ntype=ntype))
            else:                 # `"silhouette" in filepath` decides the scaling (:311-314)
                specs.append(ModalitySpec("silhouette" if "silhouette" in datadir[m] else "gray", 1))
        self.dim, self.n_classes, self.batch_size = dim, n_classes, batch_size
        kept, kept_gait = self._drop_empty(allSamples, gait, datadir)          # __remove_empty_files (:118-146)
        super().__init__(kept, kept_gait, datadir, specs, batch_size, n_classes, labmap=labmap or None, expand_level=expand_level,
                         repetition=repetition, shuffle=shuffle, is_test=isTest, keep_data=keep_data, single_input=nmods == 1)
        self.allSamples, self.gait = kept, kept_gait

    @staticmethod
    def _compress_factor(all_samples, datadir, m):
        for files, _ in all_samples:
            if files[m] != -1 and os.path.exists(os.path.join(datadir[m], files[m])):
                return float(_samples.load_sample(os.path.join(datadir[m], files[m])).get("compressFactor", 1))
        return 100.0

    @staticmethod
    def _drop_empty(all_samples, gait, datadir):
        """Keep a record when the file of its first listed modality (and of the second, if listed) exists with data."""
        def ok(m, name):
            p = os.path.join(datadir[m], name)
            return os.path.exists(p) and len(_samples.load_sample(p)["data"]) > 0
        kept, kept_gait = [], []
        for i, rec in enumerate(all_samples):
            f0 = rec[0][0]
            f1 = rec[0][1] if len(datadir) > 1 and len(rec[0]) > 1 else -1
            if f0 != -1:
                good = ok(0, f0) and (f1 == -1 or ok(1, f1))
            else:
                good = f1 != -1 and ok(1, f1)
            if good:
                kept.append(rec)
                kept_gait.append(gait[i])
        return kept, np.asarray(kept_gait)
