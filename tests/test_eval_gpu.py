"""The evaluation route of the reference (mains/mj_testUWYHGaitNet_open_tum.py:139-224, 328-341): signatures from
`Model(model.input, model.get_layer('flatten').output)` under every modality combination, then k-NN over the gallery."""
import itertools

import numpy as np
import pytest

from oracle import ugaitnet_oracle as O
from tests.synth import make_batch

pytestmark = pytest.mark.gpu


def test_signatures_for_all_seven_modality_combinations_and_knn(dev):
    from ugaitnet_amd.knn import KNeighborsClassifier
    from ugaitnet_amd.nets.mj_uwyhNets_ba import Model, UWYHSemiNet3Mods, optimizers, sign_max
    shapes = [(5, 60, 60, 2), (5, 60, 60, 1), (5, 60, 60, 1)]
    model = UWYHSemiNet3Mods.build_or_load(shapes, 4, [7, 5, 3, 2], [96, 192, 512, 4096], ndense_units=0,
                                           optimizer=optimizers.Adam(lr=1e-3), margin=0.2, nclasses=6,
                                           loss_weights=[1.0, 0.1], fMerge=sign_max, gaitset=True, seed=3)
    model_code = Model(model.input, model.get_layer("flatten").output)
    b = 8
    xs, _, labels, _ = make_batch(("of", "gray", "depth"), b, 5, 6, ids=4, seed=11)
    combos = [c for c in itertools.product((0, 1), repeat=3) if any(c)]        # the 7 test-time combinations (:599-601)
    assert len(combos) == 7
    # the model's own parameters, handed to the CPU restatement: what `predict` returns must be O.model_forward's `flat` -- the gated,
    # fused, batch-normalised signature transposed [1,0,2] and flattened (nets/mj_uwyhNets_ba.py:815-818,847) -- under EVERY combination
    p32 = model.core.get_params_numpy()
    p64 = O.cast_params(p32, np.float64)
    codes, worst = {}, 0.0
    for c in combos:
        X, us = [], []
        for m in range(3):
            u = np.full((b, 1), float(c[m]), np.float32)
            X += [xs[m], u]
            us.append(u.astype(np.float64))
        codes[c] = model_code.predict(X)
        assert codes[c].shape == (b, 62 * 256) and np.isfinite(codes[c]).all()
        r = O.model_forward([x.astype(np.float64) for x in xs], us, p64, mode="sign_max")
        err = float(np.abs(codes[c] - r["flat"]).max())
        worst = max(worst, err)
        # north_star: signatures within 1e-3.  Measured over the seven combinations on the default fp32-tensor path: 1e-5 ... 3.3e-5 (the
        # batch-axis normalisation over eight clips amplifies a small column's rounding; the golden fixtures measure 3.6e-5): bar 1e-4
        assert err <= 1e-4, (c, err)
        # the selected modality of every element is the oracle's, except where two candidates tie to fp32 rounding
        sel = model.core.sel.cpu().numpy()
        diff = sel != r["sel"]
        if diff.any():
            g = np.stack([np.abs(o * u.reshape(1, -1, 1)) for o, u in zip(r["outs"], us)])       # [3, 62, b, 256]
            top = np.sort(g, axis=0)
            assert ((top[-1] - top[-2])[diff] <= 8 * 1.2e-7 * top[-1].max()).all(), (c, int(diff.sum()))
        assert diff.mean() < 1e-4
    print("flatten codes vs the oracle under the 7 modality combinations: max |error| %.2e" % worst)
    # a disabled modality's pixels never matter; an enabled one does
    X = []
    for m, flag in enumerate((1, 0, 1)):
        X += [xs[m] if flag else np.full_like(xs[m], 7.0), np.full((b, 1), float(flag), np.float32)]
    assert np.array_equal(model_code.predict(X), codes[(1, 0, 1)])
    assert not np.array_equal(codes[(1, 0, 1)], codes[(1, 1, 1)])
    # gallery = all-modality codes, probes = the same clips: every probe's nearest neighbour is itself
    clf = KNeighborsClassifier(n_neighbors=1).fit(codes[(1, 1, 1)], labels.astype(np.float64))
    nbr, pred = clf.kneighbors_and_predict(codes[(1, 1, 1)])
    assert np.array_equal(nbr[:, 0], np.arange(b)) and np.array_equal(pred, labels.astype(np.float64))
    # probes with a modality missing are classified by the same call the reference makes
    pred2 = KNeighborsClassifier(n_neighbors=3).fit(codes[(1, 1, 1)], labels).predict(codes[(1, 1, 0)])
    assert pred2.shape == (b,) and set(pred2.tolist()) <= set(labels.tolist())
