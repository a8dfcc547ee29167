"""BASELINE.json's full sizes (C3: 3 modalities, B=24, L=25, 150 classes) checked through size-independent properties
plus an oracle spot check on two of the 24 clips (encoders are per-clip independent before the batch-axis norm)."""
import numpy as np
import pytest
import torch

from oracle import ugaitnet_oracle as O
from tests.synth import make_batch

pytestmark = pytest.mark.gpu

KINDS, B, L, NCLS = ("of", "gray", "depth"), 24, 25, 150


@pytest.fixture(scope="module")
def setup(dev):
    from ugaitnet_amd.engine import GaitCore
    xs, uses, labels, onehot = make_batch(KINDS, B, L, NCLS, seed=232323)
    core = GaitCore([2, 1, 1], nclasses=NCLS, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=232323)
    return core, xs, uses, labels, onehot


def test_encoder_outputs_of_two_clips_match_oracle(setup):
    core, xs, uses, labels, onehot = setup
    core.forward(xs, uses)
    p = core.get_params_numpy()
    for mi in (0, 1):
        rows = [int(r) for r in np.nonzero(uses[mi][:, 0] == 1)[0][[0, -1]]]   # first and last unmasked clip
        ref, _ = O.branch_forward(xs[mi][rows].astype(np.float64), {k: v.astype(np.float64) for k, v in p['branches'][mi].items()})
        got = core.encoders[mi].act['out'].cpu().numpy()[:, rows, :]
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_frame_order_does_not_matter_bit_exact(setup):
    """Set pooling is a max over the L frames: shuffling the frames of every clip leaves the signature bit-identical."""
    core, xs, uses, labels, onehot = setup
    sig = core.forward(xs, uses).cpu().numpy().copy()
    perm = np.random.default_rng(0).permutation(L)
    sig2 = core.forward([x[:, perm] for x in xs], uses).cpu().numpy()
    assert np.array_equal(sig, sig2)


def test_masked_modalities_contribute_nothing_bit_exact(setup):
    """What a masked (clip, modality) pair holds never reaches the signature: its gate factor is 0.  Bit for bit on the fp32
    path.  On the H2 path (the default) a tensor's block exponent follows the largest magnitude of the WHOLE tensor, masked clips
    included: as long as the masked clips stay below the unmasked maximum (0.3 < max|gray| = 0.5) nothing changes, bit for bit;
    a value above it (0.77) can move an exponent, which re-rounds elements below 2^-18 of the tensor's bound at the 2^-40 level:
    the signature then agrees to fp32 rounding, not to the bit."""
    core, xs, uses, labels, onehot = setup
    sig = core.forward(xs, uses).cpu().numpy().copy()
    for value in (0.3, 0.77):
        xs2 = [x.copy() for x in xs]
        for m in range(3):
            xs2[m][uses[m][:, 0] == 0] = value
        sig2 = core.forward(xs2, uses).cpu().numpy()
        if value < 0.5 or not core.h2:
            assert np.array_equal(sig, sig2), value
        else:
            assert np.abs(sig - sig2).max() <= 2e-6 * np.abs(sig).max(), np.abs(sig - sig2).max()   # (1e-3 is north_star's bar)


def test_signature_columns_have_unit_batch_norm_and_step_is_deterministic(setup):
    core, xs, uses, labels, onehot = setup
    core.forward_backward(xs, uses, labels, onehot)
    sig = core.sig.cpu().numpy()
    assert np.allclose((sig.astype(np.float64) ** 2).sum(axis=1), 1.0, atol=1e-5)
    g1 = core.store.grad.clone()
    l1 = core.losses()
    core.forward_backward(xs, uses, labels, onehot)
    assert torch.equal(g1, core.store.grad) and l1 == core.losses()      # no atomics anywhere: bitwise repeatable
    assert np.isfinite(l1['loss']) and 0 < core.bin_num.cpu().numpy().max() <= 24 * 2 * 22


def test_gradient_is_linear_in_the_loss_weights(setup):
    core, xs, uses, labels, onehot = setup
    core.loss_weights = (1.0, 0.0)
    core.forward_backward(xs, uses, labels, onehot)
    g_tri = core.store.grad.clone()
    core.loss_weights = (0.0, 1.0)
    core.forward_backward(xs, uses, labels, onehot)
    g_id = core.store.grad.clone()
    core.loss_weights = (1.0, 0.1)
    core.forward_backward(xs, uses, labels, onehot)
    ref = g_tri + 0.1 * g_id
    err = (core.store.grad - ref).norm() / ref.norm()
    assert float(err) < 1e-5


def test_skipping_masked_pairs_changes_no_result(setup):
    """skip_masked runs each encoder only on the clips whose flag is 1: the gate multiplies the rest by 0 anyway."""
    from ugaitnet_amd.engine import GaitCore
    core, xs, uses, labels, onehot = setup
    core.loss_weights = (1.0, 0.1)
    core.forward_backward(xs, uses, labels, onehot)
    sig = core.sig.cpu().numpy().copy()
    g_dense = core.store.grad.clone()
    l_dense = core.losses()
    skip = GaitCore([2, 1, 1], nclasses=NCLS, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), skip_masked=True)
    skip.set_params_numpy(core.get_params_numpy())
    skip.forward_backward(xs, uses, labels, onehot)
    assert np.array_equal(np.abs(sig), np.abs(skip.sig.cpu().numpy()))          # bit-identical up to the sign of zeros
    assert skip.encoders[0].shape[0] == int(uses[0].sum()) < 24                  # really ran on the active clips only
    ls = skip.losses()
    assert abs(ls['loss'] - l_dense['loss']) <= 1e-6 and ls['acc'] == l_dense['acc']
    err = (skip.store.grad - g_dense).norm() / g_dense.norm()
    assert float(err) < 1e-5      # weight-gradient partial sums are grouped differently over fewer frames
