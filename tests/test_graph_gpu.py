"""The training step captured as HIP graphs (engine.GraphedTrainStep) against the eager step: bit-identical parameters."""
import numpy as np
import pytest
import torch

from tests.synth import make_batch

pytestmark = pytest.mark.gpu


def _core(seed=3, **kw):
    from ugaitnet_amd.engine import GaitCore
    return GaitCore([2, 1, 1], nclasses=6, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=seed, lr=1e-3, **kw)


@pytest.mark.parametrize("prec", ["f32", "bf16", "h2"])
def test_graphed_step_equals_eager_step(dev, prec):
    from ugaitnet_amd.engine import GraphedTrainStep
    kinds = ("of", "gray", "depth")
    batches = [make_batch(kinds, 6, 3, 6, ids=3, seed=50 + i) for i in range(4)]
    eager, graphed = _core(conv_precision=prec), _core(conv_precision=prec)
    g = GraphedTrainStep(graphed, *batches[0])
    assert np.array_equal(eager.store.flat.cpu().numpy(), graphed.store.flat.cpu().numpy())     # capture updates nothing
    for xs, uses, labels, onehot in batches:
        labels = labels + 2      # other identities, same equality structure
        oh = np.eye(6, dtype=np.float32)[labels % 6]
        eager.train_step(xs, uses, labels, oh)
        g.step(xs, uses, labels, oh)
        assert eager.losses() == graphed.losses()
    torch.cuda.synchronize()
    assert np.array_equal(eager.store.flat.cpu().numpy(), graphed.store.flat.cpu().numpy())
    assert eager.iterations == graphed.iterations == 4
    with pytest.raises(ValueError):      # another label structure needs another capture
        g.step(*make_batch(kinds, 6, 3, 6, ids=2, seed=9))
