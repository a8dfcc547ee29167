"""Training-batch source that keeps the pixels off the host path: the reference generator's label-cycling
sampler + its sample files + the device-side assembly of ugaitnet_amd/batching.py.

`LabelCyclingSampler` restates how `DataGeneratorGaitMMUWYH` of data/mj_dataGeneratorMMUWYHsingle_repetitions.py picks the
records of a batch: `__prepare_labels` / `__prepare_gaits` (:222-251) index the records by (gait type, label); every epoch
(`on_epoch_end`, :254-271) resets one read pointer per (gait type, label) and shuffles the label order; `__getitem__`
(:149-183) then walks the gait types round-robin for the current label, takes the next record of that (gait type, label)
cell, and moves on to the next label after `repetition` pairs of visits -- so a batch of 2*repetition*P rows holds P labels
with 2*repetition records each (the P x K batch the batch-all triplet loss wants, 4 x 10 for CASIA-B).

`DeviceDataGenerator` is the `keras.utils.Sequence`-like object `model.fit` takes: it reads the sample files of the chosen
records (ugaitnet_amd/samples.py), uploads their raw int16 / uint8 arrays once and lets `DeviceBatchAssembler` build the
expanded, masked fp32 batch in HBM; labels follow `__gen_batchMM` (:764-780, 806-812): `labmap[label]` repeated for every
expanded row, one-hot beside it.  Augmentation (random affine transforms, :706-728) is not implemented: the generator is the
reference's with `augmentation_x=0`.
"""
from __future__ import annotations

import os
import random as _random

import numpy as np

from . import samples as _samples
from .batching import DeviceBatchAssembler, plan_rows, plan_rows_2mod


class LabelCyclingSampler:
    def __init__(self, targets, gaits, batch_size, repetition=4, shuffle=True, is_test=False, rng=np.random):
        """targets[i] / gaits[i]: label and gait type of record i.  rng: numpy-style module or Generator offering shuffle()
        (the reference uses the global np.random)."""
        self.targets = [int(t) for t in targets]
        self.gaits = np.asarray(gaits)
        if len(self.targets) != len(self.gaits):
            raise ValueError("one gait type per record")
        self.batch_size, self.repetition = int(batch_size), int(repetition)
        self.shuffle, self.is_test, self.rng = bool(shuffle), bool(is_test), rng
        self.ulabs = np.unique(self.targets)                                       # :244
        t = np.array(self.targets)
        self.lab2rec = {int(l): np.where(t == l)[0].tolist() for l in self.ulabs}  # :246-251
        self.ugait = np.unique(self.gaits)                                         # :223
        self.gait2idx = {}
        for g in self.ugait:                                                       # :227-238
            idx_g = np.where(self.gaits == g)[0]
            sub = t[idx_g]
            self.gait2idx[g] = {int(l): [int(idx_g[j]) for j in np.where(sub == l)[0]] for l in self.ulabs}
        self.on_epoch_end()

    def __len__(self):
        return len(self.targets) // self.batch_size                               # :113-116

    def on_epoch_end(self):
        self.gait2ptr = {g: {int(l): 0 for l in self.ulabs} for g in self.ugait}
        self.nextlab_idx = 0
        if not self.is_test and self.shuffle:
            self.rng.shuffle(self.ulabs)
            for k in self.lab2rec:       # (the reference shuffles these lists too; the sampler never reads them -- kept so
                self.rng.shuffle(self.lab2rec[k])   # that a seeded generator is consumed identically)

    def next_batch(self):
        if not any(len(v) for cells in self.gait2idx.values() for v in cells.values()):
            raise ValueError("no records")
        out = []
        used = used_rep = 0
        while len(out) < self.batch_size:
            for g in self.ugait:
                if len(out) == self.batch_size:
                    continue
                lab = int(self.ulabs[self.nextlab_idx])
                ptrs, recs = self.gait2ptr[g], self.gait2idx[g][lab]
                if recs:
                    out.append(recs[ptrs[lab]])
                used += 1
                ptrs[lab] += 1
                if ptrs[lab] >= len(recs):
                    ptrs[lab] = 0
                if used >= 2:
                    used = 0
                    used_rep += 1
                    if used_rep == self.repetition:
                        self.nextlab_idx += 1
                        used_rep = 0
                        if self.nextlab_idx >= len(self.ulabs):
                            self.nextlab_idx = 0
        return out


class DeviceDataGenerator:
    """all_samples[i] = ((file of modality 0 | -1, file of modality 1 | -1, ...), label) as in the reference's lists;
    datadirs[m]: directory of modality m; specs[m]: batching.ModalitySpec.  `__getitem__` returns (X, y) with X = [x_0, use_0,
    x_1, use_1, ...] device tensors and y = [labels [rows,1], one-hot [rows,n_classes]] (labels alone when n_classes == 0)."""

    def __init__(self, all_samples, gaits, datadirs, specs, batch_size, n_classes, labmap=None, expand_level=2, repetition=4,
                 shuffle=True, is_test=False, keep_data=False, rng=np.random, mask_rng=_random, device=None, single_input=False):
        self.all_samples = list(all_samples)
        self.datadirs, self.specs = list(datadirs), list(specs)
        if len(self.datadirs) != len(self.specs):
            raise ValueError("one data directory per modality")
        self.n_classes, self.labmap, self.expand = int(n_classes), labmap, max(1, int(expand_level))
        self.sampler = LabelCyclingSampler([s[1] for s in self.all_samples], gaits, batch_size, repetition, shuffle, is_test, rng)
        self.assembler = DeviceBatchAssembler(self.specs, device)
        self.mask_rng = mask_rng
        # single-modality graph (`__gen_batchSingle`, :548-656): X is the tensor itself, no flags, no expansion
        self.single_input = bool(single_input)
        if self.single_input:
            if len(self.specs) != 1:
                raise ValueError("single_input needs exactly one modality")
            self.expand = 1
        self.cache = {} if keep_data else None

    def __len__(self):
        return len(self.sampler)

    def on_epoch_end(self):
        self.sampler.on_epoch_end()

    def _data(self, m, name):
        key = (m, name)
        if self.cache is not None and key in self.cache:
            return self.cache[key]
        a = np.asarray(_samples.load_sample(os.path.join(self.datadirs[m], name))["data"])
        want = (60, 60, 25 * self.specs[m].channels)
        if a.shape != want:     # (never rely on numpy broadcasting a short or empty array into the batch buffer)
            raise ValueError("%s: data of shape %r, expected %r for modality %d" % (name, a.shape, want, m))
        if self.cache is not None:
            self.cache[key] = a
        return a

    def __getitem__(self, index):
        ids = self.sampler.next_batch()          # (like the reference, the batch does not depend on `index`)
        nmod = len(self.specs)
        present = np.zeros((len(ids), nmod), bool)
        raws = [np.zeros((len(ids), 60, 60, 25 * s.channels), np.int16 if s.is_int16 else np.uint8) for s in self.specs]
        labels = np.empty((len(ids) * self.expand, 1), np.float32)
        for i, rec in enumerate(ids):
            files, label = self.all_samples[rec]
            for m in range(nmod):
                if files[m] != -1 and files[m] is not None:
                    raws[m][i] = self._data(m, files[m])
                    present[i, m] = True
            lb = self.labmap[int(label)] if self.labmap else label
            labels[i * self.expand:(i + 1) * self.expand, 0] = lb
        plan = (plan_rows_2mod if nmod == 2 else plan_rows)(present, self.expand, rng=self.mask_rng)
        X = []
        for x, u in self.assembler.assemble(raws, plan, present=present):
            X += [x, u]
        if self.single_input:
            if not present.all():
                raise ValueError("single-modality batch with a missing sample file")     # (the reference stops in pdb here)
            X = X[0]
        if self.n_classes > 0:
            onehot = np.zeros((labels.shape[0], self.n_classes), np.float32)
            onehot[np.arange(labels.shape[0]), labels[:, 0].astype(np.int64)] = 1.0      # keras.utils.to_categorical (:812)
            return X, [labels, onehot]
        return X, labels
