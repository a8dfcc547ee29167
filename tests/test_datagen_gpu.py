"""Sample files -> label-cycling sampler -> device-side batch assembly -> model.fit (SURVEY 8(f) ranks 2-4 together)."""
import os
import random
import shutil

import numpy as np
import pytest

from oracle import batch_oracle as BO
from ugaitnet_amd import h5lite

pytestmark = pytest.mark.gpu

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
SPECS = [dict(compress_factor=100.0, channels=2), dict(compress_factor=1.0, channels=1), dict(compress_factor=1.0, channels=1)]


def _write_sample(path, data, label, gait):
    w = h5lite.Writer()
    w.create_dataset("data", data)
    for k, v in (("label", np.uint16(label)), ("videoId", np.uint16(7)), ("gait", np.uint8(gait)),
                 ("compressFactor", np.uint8(100 if data.dtype == np.int16 else 1))):
        w.set_attr("", k, v)
    w.save(path)


def test_generator_feeds_fit_from_sample_files(dev, tmp_path):
    from ugaitnet_amd.batching import ModalitySpec
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet3Mods, optimizers, sign_max
    from ugaitnet_amd.sampler import DeviceDataGenerator
    rng = np.random.default_rng(12)
    dirs = [str(tmp_path / m) for m in ("of", "gray", "depth")]
    for d in dirs:
        os.makedirs(d)
    all_samples, gaits, arrays = [], [], []
    labels = [3, 3, 3, 3, 8, 8, 8, 8]
    for i, lab in enumerate(labels):
        of = rng.integers(-3000, 3000, (60, 60, 50)).astype(np.int16)
        gray = rng.integers(0, 256, (60, 60, 25)).astype(np.uint8)
        depth = rng.integers(0, 256, (60, 60, 25)).astype(np.uint8)
        names = ["s%02d.h5" % i] * 3
        if i == 0:     # a file as deepdish / PyTables lay it out (chunked, shuffled, deflated), written by the HDF5 library
            shutil.copy(os.path.join(H5, "dd_sample_of.h5"), os.path.join(dirs[0], names[0]))
            of = np.load(os.path.join(H5, "dd_sample_of.npz"))["data"]
        else:
            _write_sample(os.path.join(dirs[0], names[0]), of, lab, i % 2)
        _write_sample(os.path.join(dirs[1], names[1]), gray, lab, i % 2)
        row = [of, gray, depth]
        if i == 5:     # no depth recording for this sample
            names[2] = -1
            row[2] = None
        else:
            _write_sample(os.path.join(dirs[2], names[2]), depth, lab, i % 2)
        all_samples.append((tuple(names), lab))
        gaits.append(i % 2)
        arrays.append(row)
    specs = [ModalitySpec("of", 2, compress_factor=100.0), ModalitySpec("gray", 1), ModalitySpec("depth", 1)]
    labmap = {3: 0, 8: 1}
    gen = DeviceDataGenerator(all_samples, gaits, dirs, specs, batch_size=8, n_classes=2, labmap=labmap, expand_level=2,
                              repetition=2, shuffle=False, mask_rng=random.Random(41))
    X, y = gen[0]
    ids = [0, 1, 2, 3, 4, 5, 6, 7]     # 2 labels x (2 gait types x 2 pairs): records in label order, gait types alternating
    assert y[0].reshape(-1).tolist() == [0, 0] * 4 + [1, 1] * 4 and y[1].shape == (16, 2)
    x_ref, _ = BO.gen_batch_mm([arrays[i] for i in ids], SPECS, 2, seed=41)
    for k in range(6):
        assert np.array_equal(X[k].cpu().numpy(), x_ref[k]), "input %d" % k
    model = UWYHSemiNet3Mods.build_or_load([(25, 60, 60, 2), (25, 60, 60, 1), (25, 60, 60, 1)], 4, [7, 5, 3, 2],
                                           [96, 192, 512, 4096], optimizer=optimizers.Adam(lr=1e-4), nclasses=2,
                                           loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=1)
    hist = model.fit(gen, epochs=2, steps_per_epoch=1, verbose=0)
    assert len(hist.history["loss"]) == 2 and np.isfinite(hist.history["loss"]).all()
    ck = str(tmp_path / "model-final-0002_weights.hdf5")
    model.save_weights(ck)
    assert h5lite.File(ck)["mat_mul_2"].keys() == ["MatMul_kernel[2]:0"]


def test_reference_constructor_surface(dev, tmp_path):
    """The class the mains instantiate, with their keyword arguments."""
    from ugaitnet_amd.data.mj_dataGeneratorMMUWYHsingle_repetitions import DataGeneratorGaitMMUWYH
    rng = np.random.default_rng(3)
    dirs = [str(tmp_path / m) for m in ("of25_60x60", "gray25_60x60", "silhouette25_60x60")]
    for d in dirs:
        os.makedirs(d)
    all_samples, gaits = [], []
    for i, lab in enumerate([5, 5, 9, 9]):
        name = "p%03d.h5" % i
        _write_sample(os.path.join(dirs[0], name), rng.integers(-3000, 3000, (60, 60, 50)).astype(np.int16), lab, 1)
        _write_sample(os.path.join(dirs[1], name), rng.integers(0, 256, (60, 60, 25)).astype(np.uint8), lab, 1)
        _write_sample(os.path.join(dirs[2], name), (rng.integers(0, 2, (60, 60, 25)) * 255).astype(np.uint8), lab, 1)
        all_samples.append(((name, name, name), lab))
        gaits.append(1)
    _write_sample(os.path.join(dirs[0], "empty.h5"), np.zeros((0,), np.int16), 5, 1)      # a recording without detections
    all_samples.append((("empty.h5", -1, -1), 5))
    gaits.append(1)
    # ... and the same as deepdish stores it: the SHAPE as an int64 array + the node attribute `zeroarray_dtype`
    w = h5lite.Writer()
    w.create_dataset("data", np.array([0], np.int64))
    w.set_attr("data", "zeroarray_dtype", np.bytes_(b"<i2"))
    w.set_attr("", "label", np.uint16(5))
    w.save(os.path.join(dirs[0], "empty_dd.h5"))
    all_samples.append((("empty_dd.h5", -1, -1), 5))
    gaits.append(1)
    kw = dict(batch_size=4, dim=[(50, 60, 60), (25, 60, 60), (25, 60, 60)], n_classes=2, datadir=dirs, labmap={5: 0, 9: 1},
              gait=gaits, ntype=2, augmentation_x=0, expand_level=3, nmods=3, gaitset=True, repetition=1, shuffle=False)
    gen = DataGeneratorGaitMMUWYH(all_samples, **kw)
    assert len(gen.allSamples) == 4 and len(gen) == 1                  # the empty file is dropped (:118-146)
    X, y = gen[0]
    assert [tuple(t.shape) for t in X] == [(12, 25, 60, 60, 2), (12, 1), (12, 25, 60, 60, 1), (12, 1), (12, 25, 60, 60, 1), (12, 1)]
    assert y[0].reshape(-1).tolist() == [0] * 6 + [1] * 6 and y[1].shape == (12, 2)
    sil = X[4][0].cpu().numpy()
    assert set(np.unique(sil)) <= {0.0, 1.0}                           # "silhouette" in the directory name: data / 255, no offset
    assert abs(float(X[0][0].abs().max().cpu()) - 3.0) < 0.01          # int16 / compressFactor 100 * 0.1
    with pytest.raises(NotImplementedError, match="augmentation_x"):
        DataGeneratorGaitMMUWYH(all_samples, **dict(kw, augmentation_x=1))
    with pytest.raises(NotImplementedError, match="augmentation_x"):    # the reference's default, as its gaitset main leaves it
        DataGeneratorGaitMMUWYH(all_samples, **{k: v for k, v in kw.items() if k != "augmentation_x"})
    # `dim` as the CASIA-B gaitset main passes it (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:213,465): the input shapes
    casia = DataGeneratorGaitMMUWYH(all_samples, **dict(kw, nmods=2, dim=[(25, 60, 60, 2), (25, 60, 60, 1)], datadir=dirs[:2]))
    assert [(sp.kind, sp.channels) for sp in casia.specs] == [("of", 2), ("gray", 1)]
    Xc, _ = casia[0]
    assert [tuple(t.shape) for t in Xc] == [(12, 25, 60, 60, 2), (12, 1), (12, 25, 60, 60, 1), (12, 1)]
    two = DataGeneratorGaitMMUWYH(all_samples, **dict(kw, nmods=2, dim=kw["dim"][:2], datadir=dirs[:2]))   # __gen_batch rules
    X2, y2 = two[0]
    assert [tuple(t.shape) for t in X2] == [(12, 25, 60, 60, 2), (12, 1), (12, 25, 60, 60, 1), (12, 1)]
    u = np.concatenate([X2[1].cpu().numpy(), X2[3].cpu().numpy()], axis=1).reshape(4, 3, 2)
    assert (u[:, 0] == 1).all() and (u[:, 1].sum(axis=1) == 1).all() and (u[:, 1] + u[:, 2] == 1).all()   # one off, then the other
    one = DataGeneratorGaitMMUWYH(all_samples, **dict(kw, nmods=1, dim=(25, 60, 60), datadir=dirs[1:2] + dirs[:1]))
    assert one.specs[0].kind == "gray"
    X1, y1 = one[0]                                                    # __gen_batchSingle: the tensor itself, no expansion
    assert tuple(X1.shape) == (4, 25, 60, 60, 1) and y1[0].reshape(-1).tolist() == [0, 0, 1, 1]
    assert float(X1.min().cpu()) >= -0.5 and float(X1.max().cpu()) <= 0.5
    with pytest.raises(NotImplementedError):
        DataGeneratorGaitMMUWYH(all_samples, **dict(kw, gaitset=False))
