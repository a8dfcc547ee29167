"""The numpy oracle (hand-derived backward) against the independent torch-autograd statement, in fp64."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T
from oracle import ugaitnet_oracle as O
from tests.synth import make_batch


def rel(a, b):
    b = b.detach().numpy()
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


@pytest.mark.parametrize("mode,kinds", [("sign_max", ("of", "gray", "depth")), ("max", ("of", "gray")), ("avg", ("of", "gray"))])
def test_multimodal_forward_backward_fp64(mode, kinds):
    b, l, ncls = 4, 2, 5
    xs, uses, labels, onehot = make_batch(kinds, b, l, ncls, ids=2, seed=21, dtype=np.float64)
    rng = np.random.default_rng(3)
    p = dict(branches=[O.init_branch_params(rng, 2 if k == 'of' else 1, np.float64) for k in kinds],
             head=O.init_head_params(rng, ncls, np.float64))
    r, g = O.model_loss_and_grads(xs, uses, labels, onehot, p, mode=mode)
    tp = T.params_from_numpy(p, torch.float64)
    tr, tg = T.loss_and_grads([torch.tensor(x) for x in xs], [torch.tensor(u) for u in uses], torch.tensor(labels),
                              torch.tensor(onehot), tp, mode=mode)
    assert rel(r['signature'], tr['signature']) < 1e-11
    assert rel(r['logits'], tr['logits']) < 1e-11
    assert abs(float(r['loss']) - float(tr['loss'])) < 1e-11
    for i in range(len(kinds)):
        for k in g['branches'][i]:
            assert rel(g['branches'][i][k], tg['branches'][i][k]) < 1e-9, (i, k)
    for k in g['head']:
        assert rel(g['head'][k], tg['head'][k]) < 1e-9, k


def test_single_modality_fp64():
    b, l, ncls = 4, 2, 5
    xs, _, labels, onehot = make_batch(('gray',), b, l, ncls, ids=2, seed=22, dtype=np.float64)
    rng = np.random.default_rng(4)
    p = dict(branches=[O.init_branch_params(rng, 1, np.float64)], head=O.init_head_params(rng, ncls, np.float64))
    r, g = O.model_loss_and_grads(xs, None, labels, onehot, p, multimodal=False)
    tp = T.params_from_numpy(p, torch.float64)
    tr, tg = T.loss_and_grads([torch.tensor(xs[0])], None, torch.tensor(labels), torch.tensor(onehot), tp, multimodal=False)
    assert rel(r['signature'], tr['signature']) < 1e-11
    # batch_dist's diagonal is x2_i + x2_i - 2 x_i.x_i: rounding noise, not 0, so d_ii ~ 1e-9 and its 1/(2 d) factor
    # amplifies the (analytically cancelling) diagonal terms of the backward; both implementations carry that noise
    # (as the reference does), hence 1e-6 here instead of 1e-9.  The HIP kernel takes the norms from the Gram diagonal,
    # so its d_ii is exactly 0.
    for k in g['branches'][0]:
        assert rel(g['branches'][0][k], tg['branches'][0][k]) < 1e-6, k


def test_fp32_forward_close_to_fp64():
    kinds, b, l, ncls = ("of", "gray", "depth"), 4, 2, 5
    xs, uses, labels, onehot = make_batch(kinds, b, l, ncls, ids=2, seed=23)
    rng = np.random.default_rng(5)
    p64 = dict(branches=[O.init_branch_params(rng, 2 if k == 'of' else 1, np.float64) for k in kinds],
               head=O.init_head_params(rng, ncls, np.float64))
    r64 = O.model_forward([x.astype(np.float64) for x in xs], [u.astype(np.float64) for u in uses], p64)
    r32 = O.model_forward(xs, uses, O.cast_params(p64, np.float32))
    assert np.abs(r32['signature'] - r64['signature']).max() < 2e-5
