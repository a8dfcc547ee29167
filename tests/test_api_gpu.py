"""The drop-in boundary: the reference's class API (nets/mj_uwyhNets_ba.py) and the Keras subset its mains use."""
import os

import numpy as np
import pytest

from tests.synth import make_batch

pytestmark = pytest.mark.gpu


class ToyGenerator:
    """keras.utils.Sequence-shaped generator with the reference's batch contract:
    X = [of, use_of, gray, use_gray, depth, use_depth], y = [labels [B,1], onehot]."""

    def __init__(self, kinds, b, l, ncls, n_batches=3, multimodal=True):
        self.batches = []
        for i in range(n_batches):
            xs, uses, labels, onehot = make_batch(kinds, b, l, ncls, ids=b // 2, seed=100 + i)
            X = [a for pair in zip(xs, uses) for a in pair] if multimodal else xs[0]
            self.batches.append((X, [labels.reshape(-1, 1).astype(np.float32), onehot]))
        self.epochs_ended = 0

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        return self.batches[i]

    def on_epoch_end(self):
        self.epochs_ended += 1


def test_three_mod_build_fit_predict_save_load(dev, tmp_path):
    from ugaitnet_amd.nets.mj_uwyhNets_ba import Model, UWYHSemiNet, UWYHSemiNet3Mods, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1), (3, 60, 60, 1)]
    model = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], ndense_units=0,
                                           optimizer=optimizers.Adam(lr=1e-3), margin=0.2, nclasses=6,
                                           loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=1)
    assert [t.name for t in model.input] == ["ofinput1", "ofuse1", "grayinput1", "grayuse1", "depthinput1", "depthuse1"]
    assert model.get_layer("classprob").units == 6 and model.get_layer("fusion") is not None
    gen = ToyGenerator(("of", "gray", "depth"), 4, 3, 6)
    seen = []

    class CB:
        def on_epoch_end(self, epoch, logs):
            seen.append((epoch, sorted(logs)))

    model, hist = UWYHSemiNet.fit_generator(model, 3, [CB()], gen, gen, 1, 2, 1, new_lr=5e-4)
    assert hist.epoch == [1, 2] and gen.epochs_ended == 2            # initial_epoch=1 .. epochs=3
    assert model.optimizer.lr == 5e-4 and hist.history["lr"] == [5e-4, 5e-4]
    for k in ("loss", "signature_loss", "classprob_loss", "classprob_acc", "val_loss", "val_classprob_acc"):
        assert k in hist.history and len(hist.history[k]) == 2
    assert seen[0][0] == 1
    X, y = gen[0]
    sig, probs = model.predict(X)
    assert sig.shape == (62, 4, 256) and probs.shape == (4, 6) and np.allclose(probs.sum(1), 1, atol=1e-5)
    flat = Model(model.input, model.get_layer("flatten").output).predict(X)
    assert flat.shape == (4, 62 * 256) and np.array_equal(flat[1, 256:512], sig[1, 1])   # index k*256+d
    # masked modality rows contribute exactly nothing: zeroing their (already ignored) pixels changes no output
    X2 = [a.copy() for a in X]
    use_of = X[1].reshape(-1)
    X2[0][use_of == 0] = 123.0
    sig2, _ = model.predict(X2)
    assert np.array_equal(sig, sig2)
    # save / load round trip (full model and weights-only), classprob surgery by name with skip_mismatch
    path = os.path.join(tmp_path, "model-state-0002.hdf5")
    model.save(path)
    model.save_weights(UWYHSemiNet.get_weights_filename(path))
    again = UWYHSemiNet3Mods.loadnet(path)
    assert np.array_equal(again.predict(X)[0], sig)
    other = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], optimizer=optimizers.Adam(lr=1e-3),
                                           nclasses=9, loss_weights=[1.0, 0.1], initnet=path, fMerge=sign_max, gaitset=True)
    assert other.get_layer("classprob").units == 9
    assert np.array_equal(other.predict(X)[0], sig)                   # encoders were loaded, head re-initialised
    w = model.get_layer("ofBranch_a1").get_weights()[0]
    assert w.shape == (5, 5, 2, 32)


def test_freeze_all_trains_the_classifier_only(dev, tmp_path):
    """build_or_load(initnet=..., freeze_all=True) on the gaitset path (reference :635-649): a fresh model takes the weights of
    `<initnet>_weights.hdf5` by name and every layer but `classprob` stops training; freeze_convs alone freezes nothing."""
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet, UWYHSemiNet3Mods, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1), (3, 60, 60, 1)]
    kw = dict(optimizer=optimizers.Adam(lr=1e-2), nclasses=6, loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True)
    base = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], seed=1, **kw)
    path = os.path.join(tmp_path, "model-state-0002.hdf5")
    base.save(path)
    base.save_weights(UWYHSemiNet.get_weights_filename(path))
    gen = ToyGenerator(("of", "gray", "depth"), 4, 3, 6)
    X, y = gen[0]
    frozen = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], initnet=path, freeze_all=True, **kw)
    before = frozen.core.store.flat.cpu().numpy().copy()
    for _ in range(3):
        frozen.train_on_batch(X, y)
    after = frozen.core.store.flat.cpu().numpy()
    lo, hi = frozen.core._buckets[3]                      # the head's slice of the flat parameter buffer
    assert np.array_equal(before[:lo], after[:lo]) and not np.array_equal(before[lo:hi], after[lo:hi])
    assert np.array_equal(frozen.predict(X)[0], base.predict(X)[0])      # signatures: the encoders are the loaded ones
    # the head's update is the one the unfrozen model makes to its head on the first step
    free = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], initnet=path, freeze_convs=True, **kw)
    f0 = free.core.store.flat.cpu().numpy().copy()
    free.train_on_batch(X, y)
    f1 = free.core.store.flat.cpu().numpy()
    assert not np.array_equal(f0[:lo], f1[:lo])            # freeze_convs alone: everything still trains on the gaitset path
    again = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], initnet=path, freeze_all=True, **kw)
    again.train_on_batch(X, y)
    assert np.array_equal(again.core.store.flat.cpu().numpy()[lo:hi], f1[lo:hi])


def test_keras_surface_skips_masked_pairs_without_changing_results(dev):
    """Models built through the reference's class API run each encoder only on the clips whose modality flag is 1."""
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet3Mods, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1), (3, 60, 60, 1)]
    mk = lambda: UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], optimizer=optimizers.Adam(lr=1e-3),
                                                nclasses=6, loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=4)
    a, b = mk(), mk()
    assert a.core.skip_masked
    b.core.skip_masked = False
    X, y = ToyGenerator(("of", "gray", "depth"), 8, 3, 6, n_batches=1)[0]      # 8 rows: every mask pattern at least once
    pa, pb = a.predict(X), b.predict(X)
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
    la, lb = a.train_on_batch(X, y), b.train_on_batch(X, y)
    assert abs(la["loss"] - lb["loss"]) <= 1e-6
    wa, wb = a.core.store.flat.cpu().numpy(), b.core.store.flat.cpu().numpy()
    assert np.abs(wa - wb).max() <= 2e-4        # (Adam's first step moves an element by up to lr = 1e-3; the two gradients differ by rounding)


def test_single_and_two_modality_models_and_encode(dev):
    from ugaitnet_amd.nets.mj_uwyhNets_ba import Maximum, UWYHSemiNet, optimizers
    one = UWYHSemiNet.build_or_load((3, 60, 60, 1), 4, [7, 5, 3, 2], [96, 192, 512, 4096], optimizer=optimizers.Adam(lr=1e-4),
                                    nclasses=4, loss_weights=[1.0, 0.1], gaitset=True, seed=2)
    gen = ToyGenerator(("gray",), 4, 3, 4, multimodal=False)
    hist = one.fit(gen, epochs=1, steps_per_epoch=2, verbose=0)
    assert "mat_mul_loss" in hist.history
    two = UWYHSemiNet.build_or_load([(3, 60, 60, 2), (3, 60, 60, 1)], 4, [7, 5, 3, 2], [96, 192, 512, 4096],
                                    optimizer=optimizers.Adam(lr=1e-4), nclasses=0, fMerge=Maximum, gaitset=True, seed=3)
    xs, uses, labels, _ = make_batch(("of", "gray"), 4, 3, 4, ids=2, seed=5)
    logs = two.train_on_batch([xs[0], uses[0], xs[1], uses[1]], labels.reshape(-1, 1))
    assert set(logs) == {"loss"} and np.isfinite(logs["loss"])
    codes = UWYHSemiNet.encode(two, [xs[0], xs[1]], uses, gaitset=True)
    assert codes.shape == (62, 4, 256)
    assert np.allclose((codes ** 2).sum(axis=1), 1.0, atol=1e-4)      # l2_normalize over the batch axis


def test_unsupported_configurations_fail_loudly(dev):
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet, optimizers
    with pytest.raises(NotImplementedError):
        UWYHSemiNet.build((3, 60, 60, 1), 4, [7], [96], optimizer=optimizers.SGD(0.001, 0.9), gaitset=True)
    with pytest.raises(ValueError):
        UWYHSemiNet.build((3, 50, 60, 1), 4, [7], [96], optimizer=optimizers.Adam(), gaitset=True)


def test_checkpoints_are_keras_hdf5_files(dev, tmp_path):
    """save_weights / save write the Keras HDF5 layout (ugaitnet_amd/keras_h5.py); load_weights restores from it by the
    reference's layer names, takes the legacy npz container too, and honours by_name / skip_mismatch."""
    from ugaitnet_amd import h5lite, keras_h5
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1)]
    mk = lambda seed, ncls: UWYHSemiNet.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], nclasses=ncls,
                                                      optimizer=optimizers.Adam(lr=1e-3), loss_weights=[1.0, 0.1],
                                                      fMerge=sign_max, gaitset=True, seed=seed)
    a, b = mk(1, 5), mk(2, 5)
    xs, uses, labels, onehot = make_batch(("of", "gray"), 4, 3, 5, ids=2, seed=8)
    X, y = [xs[0], uses[0], xs[1], uses[1]], [labels.reshape(-1, 1), onehot]
    a.train_on_batch(X, y)
    wpath = os.path.join(tmp_path, "model-final-0001_weights.hdf5")
    a.save_weights(wpath)
    f = h5lite.File(wpath)
    names = [n.decode() for n in f.attrs["layer_names"]]
    assert names[:4] == ["time_distributed_1", "time_distributed_3", "conv2d_2", "conv2d_3"] and names[-1] == "classprob"
    assert "time_distributed_16" in names and "mat_mul_1" in names            # second modality's counters
    assert f["time_distributed_1/time_distributed_1/kernel:0"].shape == (5, 5, 2, 32)
    assert f["classprob/classprob/kernel:0"].shape == (15872, 5)
    sa = a.predict(X)
    assert not np.array_equal(b.predict(X)[0], sa[0])
    b.load_weights(wpath, by_name=True)
    sb = b.predict(X)
    assert np.array_equal(sb[0], sa[0]) and np.array_equal(sb[1], sa[1])
    # classifier of another width: refused without skip_mismatch, encoders taken with it
    c = mk(3, 7)
    with pytest.raises(ValueError):
        c.load_weights(wpath, by_name=True)
    c2 = mk(3, 7)
    c2.load_weights(wpath, by_name=True, skip_mismatch=True)
    assert np.array_equal(c2.predict(X)[0], sa[0])
    # full model: optimizer state and step counter survive, training continues identically
    mpath = os.path.join(tmp_path, "model-state-0001.hdf5")
    a.save(mpath)
    a2 = UWYHSemiNet.loadnet(mpath)
    assert a2.core.iterations == a.core.iterations == 1
    la, la2 = a.train_on_batch(X, y), a2.train_on_batch(X, y)
    assert la == la2 and np.array_equal(a.core.store.flat.cpu().numpy(), a2.core.store.flat.cpu().numpy())
    # the npz container of earlier checkpoints still loads
    npz = os.path.join(tmp_path, "old.npz")
    a.save_weights(npz)
    d = mk(4, 5)
    d.load_weights(npz)
    assert np.array_equal(d.predict(X)[0], a.predict(X)[0])
    # a weights file is complete for by_name=False; an encoder-only file is not
    e = mk(5, 5)
    e.load_weights(wpath)
    layers = [l for l in keras_h5.read_layers(wpath) if l[0] != "classprob"]
    assert len(layers) == 22


def test_model_config_file_drives_loadnet_and_surgery(dev, tmp_path):
    """`model-config.hdf5` beside a checkpoint (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:474-489): the surgery route of
    build_or_load rebuilds from it (reference :610-630) and loadnet falls back to it for a model file it cannot interpret
    (:1008-1030), taking the arrays of `<net>_weights.hdf5` by name."""
    from ugaitnet_amd import ddconfig, h5lite
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1)]
    a = UWYHSemiNet.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], nclasses=5, margin=0.25,
                                  optimizer=optimizers.Adam(lr=1e-3), loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True,
                                  seed=3)
    xs, uses, labels, onehot = make_batch(("of", "gray"), 4, 3, 5, ids=2, seed=9)
    X = [xs[0], uses[0], xs[1], uses[1]]
    sig = a.predict(X)[0]
    net = os.path.join(tmp_path, "model-final.hdf5")
    a.save_weights(UWYHSemiNet.get_weights_filename(net))
    # what the main stores: names for optimizer and merge function, the shapes as a list of tuples
    ddconfig.save(UWYHSemiNet.get_netconfig_filename(net),
                  {"filters_size": [7, 5, 3, 2], "filters_numbers": [96, 192, 512, 4096], "input_shape": shapes,
                   "ndense_units": 0, "weight_decay": 1e-4, "dropout": 0.4, "optimizer": "Adam", "margin": 0.25,
                   "custom": "TripletSemiHardLoss", "nclasses": 5, "softlabel": 0, "use3D": False,
                   "loss_weights": [1.0, 0.1], "fMerge": "sign_max"})
    # (1) a model file of foreign make (here: a weights-only file under the model's name): loadnet rebuilds from the config
    w = h5lite.Writer()
    w.create_dataset("model_weights/placeholder", np.zeros(1, np.float32))
    w.save(net)
    b = UWYHSemiNet.loadnet(net)
    assert b.get_layer("classprob").units == 5 and b.margin == 0.25 and b.fmerge_mode == "sign_max"
    assert np.array_equal(b.predict(X)[0], sig)
    # (2) surgery: the checkpoint has 5 classes, the caller wants 9 -> configuration + caller's settings, weights by name
    a.save(net)
    c = UWYHSemiNet.build_or_load([(9, 9, 9, 9)], 4, [7, 5, 3, 2], [96, 192, 512, 4096], nclasses=9, margin=0.3,
                                  optimizer=optimizers.Adam(lr=2e-3), loss_weights=[1.0, 0.5], initnet=net, fMerge=sign_max,
                                  gaitset=True)
    assert c.get_layer("classprob").units == 9 and c.margin == 0.3 and c.loss_weights == [1.0, 0.5]
    assert [tuple(s) for s in c.input_shapes] == shapes       # from the stored configuration, not from the (wrong) argument
    assert np.array_equal(c.predict(X)[0], sig)


def test_fit_pipeline_matches_the_synchronous_loop(dev):
    """`fit` as tf.keras runs it for a Sequence (nets/mj_uwyhNets_ba.py:963: workers=1, max_queue_size=10 by default): batches fetched
    and staged in HBM by a background thread, step k's losses read after step k + 1 is queued.  Against the plain loop (workers=0,
    pipeline=False) from the same seed: the same History to the last bit, the same parameters, `on_epoch_end` of the generator
    between the epochs' fetches, a callback that listens to batch events switches the pipelining off by itself, and an exception
    inside the generator surfaces in `fit`."""
    import torch
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet3Mods, optimizers, sign_max
    shapes = [(3, 60, 60, 2), (3, 60, 60, 1), (3, 60, 60, 1)]

    def build():
        return UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], ndense_units=0,
                                              optimizer=optimizers.Adam(lr=1e-3), margin=0.2, nclasses=6,
                                              loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=3)

    class Gen(ToyGenerator):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.fetched = []

        def __getitem__(self, i):
            self.fetched.append((self.epochs_ended, i))
            return super().__getitem__(i)

    runs = {}
    for name, kw in (("plain", dict(workers=0, pipeline=False)), ("piped", dict()), ("pool", dict(workers=3, max_queue_size=2))):
        model, gen = build(), Gen(("of", "gray", "depth"), 4, 3, 6, n_batches=4)
        val = ToyGenerator(("of", "gray", "depth"), 4, 3, 6, n_batches=2)
        hist = model.fit(gen, validation_data=val, validation_steps=2, epochs=3, steps_per_epoch=4, verbose=0, **kw)
        torch.cuda.synchronize()
        runs[name] = (hist.history, {n: model.core.store.get(n).copy() for n in model.core.store.names}, gen)
    assert runs["plain"][0] == runs["piped"][0] == runs["pool"][0] and len(runs["piped"][0]["loss"]) == 3 and "val_loss" in runs["piped"][0]
    for n, w in runs["plain"][1].items():
        assert np.array_equal(w, runs["piped"][1][n]) and np.array_equal(w, runs["pool"][1][n]), n
    gen = runs["piped"][2]
    assert gen.fetched == [(e, i) for e in range(3) for i in range(4)] and gen.epochs_ended == 3      # epoch e's batches after e reshuffles
    # three fetching threads: any order inside an epoch, never across its end
    assert sorted(runs["pool"][2].fetched) == gen.fetched and [e for e, _ in runs["pool"][2].fetched] == sorted(e for e, _ in gen.fetched)

    seen = []

    class BatchCB:
        def on_batch_end(self, step, logs):
            seen.append((step, logs["loss"]))

    class Callback:                       # (what a tf.keras callback inherits: no-op hooks that must not switch the pipelining off)
        def on_batch_end(self, step, logs):
            pass

    class EpochOnly(Callback):
        def on_epoch_end(self, epoch, logs):
            seen.append(("epoch", epoch))

    from ugaitnet_amd import keras_compat
    assert keras_compat._listens(BatchCB(), "on_batch_end") and not keras_compat._listens(EpochOnly(), "on_batch_end")
    model = build()
    h2 = model.fit(Gen(("of", "gray", "depth"), 4, 3, 6, n_batches=4), epochs=1, steps_per_epoch=4, verbose=0, callbacks=[BatchCB(), EpochOnly()])
    assert [s[0] for s in seen] == [0, 1, 2, 3, "epoch"]         # batch events in order, each BEFORE the next step (no pipelining)
    assert h2.history["loss"][0] == runs["plain"][0]["loss"][0]

    class Broken(ToyGenerator):
        def __getitem__(self, i):
            if i == 2:
                raise KeyError("sample file of batch 2 is missing")
            return super().__getitem__(i)

    with pytest.raises(KeyError, match="batch 2"):
        build().fit(Broken(("of", "gray", "depth"), 4, 3, 6, n_batches=4), epochs=1, steps_per_epoch=4, verbose=0)
