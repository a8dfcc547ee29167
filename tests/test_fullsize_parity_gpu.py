"""BASELINE.json's full sizes against the oracle itself: one whole training step (forward, both losses, every parameter
gradient) of C3 (of + gray + depth, 24 clips = 12 ids x 2, 150 classes) and C4 (of + gray + silhouette, 40 clips = 4 ids x 10,
74 classes) on the default HIP path, compared with oracle/torch_ref.py evaluated in fp64 on the host's cores (the independent
torch-autograd statement that pins the numpy oracle, tests/test_oracle_crosscheck.py; the numpy oracle itself needs minutes at
these sizes).

C5 (configs[4]: C4's modalities, 16 clips per GPU, bf16 MFMA operands) runs the same comparison with the bf16 mode's bars.
Bars: loss <= 1e-4, signature <= 1e-3 (north_star's tolerance; observed ~1e-5), active-triplet counts equal up to hinges that
sit within fp32 rounding of zero (C3: exact; C4 has 744k hinges per step: at most 1 per bin, 3 in all), every parameter
gradient <= 5e-3 relative L2 (an fp32-vs-fp64 near-tie can flip a MaxPool / set-max / HPP / sign_max routing decision)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as T
from oracle import ugaitnet_oracle as O
from tests.synth import make_batch

pytestmark = pytest.mark.gpu

CASES = {
    "C3": dict(kinds=("of", "gray", "depth"), b=24, ids=12, ncls=150),
    "C4": dict(kinds=("of", "gray", "sil"), b=40, ids=4, ncls=74),
    # BASELINE.json configs[4] / SURVEY "C5": C4's modalities, 16 clips per GPU (128 per 8-GPU node, 8 ids x 16 -> 2 ids here),
    # bf16 operands on the matrix cores with fp32 accumulate
    "C5": dict(kinds=("of", "gray", "sil"), b=16, ids=2, ncls=74, precision="bf16"),
    # BASELINE.json configs[1] / SURVEY "C2": BL-single gray, 24 clips = 12 ids x 2, 150 classes -- the single-modality graph
    # (no gate, no normalisation: nets/mj_uwyhNets_ba.py:893-903) at its full size
    "C2": dict(kinds=("gray",), b=24, ids=12, ncls=150, multimodal=False),
    # the same steps with the 3x3 layers on the f16 matrix pipe (H2 tensors, ugaitnet_amd/engine_h2.py) at the fp32 bars
    "C2h2": dict(kinds=("gray",), b=24, ids=12, ncls=150, multimodal=False, precision="h2"),
    "C3h2": dict(kinds=("of", "gray", "depth"), b=24, ids=12, ncls=150, precision="h2"),
    "C4h2": dict(kinds=("of", "gray", "sil"), b=40, ids=4, ncls=74, precision="h2"),
}


def _rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5", "C2h2", "C3h2", "C4h2"])
def test_whole_step_matches_the_fp64_oracle(dev, name):
    from ugaitnet_amd.engine import GaitCore
    c = CASES[name]
    kinds, b, ncls = c["kinds"], c["b"], c["ncls"]
    xs, uses, labels, onehot = make_batch(kinds, b, 25, ncls, ids=c["ids"], seed=232323)
    rng = np.random.default_rng(11)
    p64 = dict(branches=[O.init_branch_params(rng, 2 if k == "of" else 1, np.float64) for k in kinds],
               head=O.init_head_params(rng, ncls, np.float64))
    p64["head"]["bc"] = rng.normal(size=ncls) * 0.01
    bf16 = c.get("precision") == "bf16"
    multimodal = c.get("multimodal", True)
    core = GaitCore([2 if k == "of" else 1 for k in kinds], nclasses=ncls, multimodal=multimodal, fuse_mode="sign_max", margin=0.2,
                    loss_weights=(1.0, 0.1), device=dev, conv_precision=c.get("precision", "f32"))
    core.set_params_numpy(O.cast_params(p64, np.float32))
    core.forward_backward(xs, uses if multimodal else None, labels, onehot)
    torch.cuda.synchronize()
    got = core.get_grads_numpy()
    sig = core.sig.cpu().numpy()
    ls = core.losses()
    counts = core.bin_num.cpu().numpy()

    torch.set_num_threads(max(1, torch.get_num_threads()))
    tp = T.params_from_numpy(p64, dtype=torch.float64)
    res, g = T.loss_and_grads([torch.from_numpy(x.astype(np.float64)) for x in xs],
                              [torch.from_numpy(u.astype(np.float64)) for u in uses] if multimodal else None, torch.from_numpy(labels),
                              torch.from_numpy(onehot.astype(np.float64)), tp, margin=0.2, loss_weights=(1.0, 0.1),
                              multimodal=multimodal)
    dcount = np.abs(counts.astype(np.int64) - res["tri_counts"].numpy().astype(np.int64))
    if bf16:    # bf16 operands (8 significant bits) in every 3x3 convolution: the bars of test_bf16_operand_mode_against_the_oracle
        # under sign_max a near-tie between two modalities flips the selected one (and possibly the sign) at 8 significant bits:
        # the FRACTION of such elements is bounded, the rest stays within the bf16 bar (as in test_bf16_operand_mode_...)
        assert abs(ls["loss"] - float(res["loss"])) <= 5e-2 * abs(float(res["loss"])), (ls, float(res["loss"]))
        serr = np.abs(sig - res["signature"].detach().numpy())
        assert (serr > 5e-2).mean() < 0.05 and np.median(serr) <= 5e-3, ((serr > 5e-2).mean(), np.median(serr))
        assert dcount.max() <= 0.05 * max(1.0, float(res["tri_counts"].max()))
    else:
        assert abs(ls["loss"] - float(res["loss"])) <= 1e-4, (ls, float(res["loss"]))
        assert abs(ls["triplet"] - float(res["triplet"])) <= 1e-4 and abs(ls["xent"] - float(res["xent"])) <= 1e-4
        # (the single-modality graph feeds the RAW branch output to both heads: its scale is not 1, so the bar is relative)
        sref = res["signature"].detach().numpy()
        assert np.abs(sig - sref).max() <= 1e-3 * max(1.0, float(np.abs(sref).max()))
        if name.startswith("C3") or name.startswith("C2"):
            assert dcount.max() == 0, dcount
        else:
            assert dcount.max() <= 1 and dcount.sum() <= 3, dcount
    worst = {}
    for mi in range(len(kinds)):
        for k, ref in g["branches"][mi].items():
            worst["m%d.%s" % (mi, k)] = _rell2(got["branches"][mi][k], ref.numpy())
    for k, ref in g["head"].items():
        worst["head." + k] = _rell2(got["head"][k], ref.numpy())
    # bf16 at full size (measured): 0.09 ... 0.15 on the gray / silhouette branches, 0.14 ... 0.30 on the optical-flow branch, 0.02 on
    # the classifier: decision flips of the losses and of the routing at 8 significant bits included (tests/test_engine_gpu.py
    # test_branch_gradients_with_a_fixed_cotangent separates the branches from the losses)
    bad = {k: v for k, v in worst.items() if v > ((3.5e-1 if k.startswith("m0.") else 2.5e-1) if bf16 else 5e-3)}
    if bf16:
        assert float(np.median(list(worst.values()))) <= 1.5e-1 and worst["head.wc"] <= 5e-2, worst
    assert not bad, (bad, worst)
    if bf16:
        print("%s gradient rel-L2 per tensor: %s" % (name, ", ".join("%s %.3f" % kv for kv in sorted(worst.items(), key=lambda kv: -kv[1]))))
    print("%s: loss %.6f (oracle %.6f), max |sig - oracle| %.2e, worst gradient rel-L2 %.2e (%s)"
          % (name, ls["loss"], float(res["loss"]), np.abs(sig - res["signature"].detach().numpy()).max(),
             max(worst.values()), max(worst, key=worst.get)))
