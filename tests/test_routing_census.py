"""tests/routing.py checked on the CPU: the census of an fp32 run of the torch graph against the fp64 decisions finds only near-tie
flips, the fp64 run against itself finds none, a planted wrong decision is reported as far from a tie, and forcing the oracle to its
own routing reproduces its own gradient."""
import numpy as np
import torch

from oracle import torch_ref as T
from oracle import ugaitnet_oracle as O
from tests import routing as R
from tests.synth import make_batch


def _setup():
    kinds, b, l = ("of", "gray"), 3, 4
    xs, uses, labels, onehot = make_batch(kinds, b, l, 6, ids=3, seed=3)
    rng = np.random.default_rng(9)
    ps = [O.init_branch_params(rng, 2 if k == "of" else 1, np.float64) for k in kinds]
    return kinds, b, l, xs, ps


def test_census_of_an_fp32_run_and_of_the_oracle_itself():
    kinds, b, l, xs, ps = _setup()
    for x, p in zip(xs, ps):
        x64 = torch.from_numpy(x.astype(np.float64))
        p64 = {k: torch.from_numpy(v) for k, v in p.items()}
        out, dec = R.oracle_branch_census(x64, p64)
        assert tuple(out.shape) == (62, b, 256)
        assert torch.equal(out, T.branch(x64, p64))       # the tapped graph IS the oracle's graph, statement for statement
        assert set(R.SIGN_KEYS) <= {k[3:] for k in dec if k.startswith("sg_")}
        own = R.census(dec, R.route_from_torch(x64, p64), b, l)
        assert all(f == 0 for _, f, _ in own.values()), own
        r32 = R.route_from_torch(x64.float(), {k: v.float() for k, v in p64.items()})
        res = R.census(dec, r32, b, l)
        # fp32 arithmetic decides like fp64 except at near-ties: every flip within a few hundred fp32 ulp of the tensor's scale
        assert all(w <= 512 * R.FP32_ULP for _, _, w in res.values()), R.format_census(res)
        assert sum(n for n, _, _ in res.values()) > 100000
        # a planted wrong decision far from a tie is reported as such
        bad = dict(r32)
        bad["i2"] = r32["i2"].copy()
        flat = bad["i2"].reshape(-1)
        clear = np.setdiff1d(np.arange(flat.size), _nhwc_index(dec["i2"], r32["i2"].shape))[:1]
        flat[clear] = (flat[clear] + 1) % 4
        n, f, w = R.census(dec, bad, b, l)["i2"]
        assert f >= 1 and w == np.inf


def _nhwc_index(d, nhwc_shape):
    """flat NHWC indices of the decisions of `d` (layout [N, C, H, W]) that are near-ties"""
    n, h, w, c = nhwc_shape
    i = d.near_index
    nn, cc, hh, ww = np.unravel_index(i, (n, c, h, w))
    return np.ravel_multi_index((nn, hh, ww, cc), (n, h, w, c))


def test_forcing_the_oracle_to_its_own_routing_changes_nothing():
    kinds, b, l, xs, ps = _setup()
    x64 = torch.from_numpy(xs[1].astype(np.float64))
    rng = np.random.default_rng(1)
    dout = torch.from_numpy(rng.normal(size=(62, b, 256)))
    grads = []
    for forced in (False, True):
        tp = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in ps[1].items()}
        if forced:
            route = R.route_from_torch(x64, {k: v.detach() for k, v in tp.items()})
            out = R.forced_branch(x64, tp, route)
        else:
            out = T.branch(x64, tp)
        (out * dout).sum().backward()
        grads.append({k: v.grad.numpy() for k, v in tp.items()})
    for k in grads[0]:
        assert np.abs(grads[0][k] - grads[1][k]).max() <= 1e-9 * max(1.0, np.abs(grads[0][k]).max()), k
