"""Batch-hard triplet loss (the mode `compile_hard` names, nets/mj_uwyhNets_ba.py:1301-1306): the numpy restatement with its
hand-derived backward against the torch-autograd statement of tensorflow_addons' formulas, plus hand-checkable cases."""
import numpy as np
import torch

from oracle import torch_ref as T
from oracle import ugaitnet_oracle as O


def test_hand_case():
    # one bin, 4 points on a line: identities (0,0,1,1) at x = 0, 1, 3, 7
    emb = np.array([[[0.0], [1.0], [3.0], [7.0]]])
    loss, aux = O.triplet_hard([0, 0, 1, 1], emb, 0.5)
    # anchors: hp = (1, 1, 4, 4), hn = (3, 2, 2, 6) -> hinges max(hp - hn + 0.5, 0) = (0, 0, 2.5, 0)
    assert np.allclose(aux['h'][0], [0, 0, 2.5, 0]) and np.isclose(loss, 2.5 / 4)
    # an identity with a single sample: no positive -> hp = row minimum = 0; no negatives at all -> hn = row maximum
    _, a2 = O.triplet_hard([0, 1, 1], np.array([[[0.0], [2.0], [5.0]]]), 1.0)
    assert np.allclose(a2['h'][0], [max(0 - 2 + 1, 0), max(3 - 2 + 1, 0), max(3 - 5 + 1, 0)])
    _, a3 = O.triplet_hard([4, 4], np.array([[[0.0], [2.0]]]), 1.0)
    assert np.allclose(a3['h'][0], [2 - 2 + 1, 2 - 2 + 1])


def test_numpy_backward_matches_autograd():
    rng = np.random.default_rng(5)
    for labels in ([0, 0, 1, 1, 2, 2], [0, 1, 1, 1, 2, 0, 3], [5, 5, 5]):
        m = len(labels)
        emb = rng.normal(size=(4, m, 6))
        loss, aux = O.triplet_hard(labels, emb, 0.2)
        g = O.triplet_hard_bwd(emb, aux)
        t = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
        tl = T.triplet_hard(torch.tensor(labels), t, 0.2)
        tl.backward()
        assert abs(float(tl) - loss) <= 1e-12
        assert np.abs(t.grad.numpy() - g).max() <= 1e-10
