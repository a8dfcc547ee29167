import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.synth import make_batch
from tests.test_engine_gpu import build, oracle_params, rell2
from oracle import ugaitnet_oracle as O
from ugaitnet_amd import bf16
kinds, b, l, ncls = ('of', 'gray', 'depth'), 8, 4, 10
xs, uses, labels, onehot = make_batch(kinds, b, l, ncls, ids=4, seed=1)
p64 = oracle_params(kinds, ncls)
core = build(kinds, ncls, 'avg', p64, conv_precision='bf16')
r, g = O.model_loss_and_grads([x.astype(np.float64) for x in xs], [u.astype(np.float64) for u in uses], labels, onehot.astype(np.float64), p64, margin=0.2, loss_weights=(1.0, 0.1), mode='avg')
core.forward_backward(xs, uses, labels, onehot)
torch.cuda.synchronize()
got = core.get_grads_numpy()
for mi in range(3):
    print(mi, {k: "%.3g" % rell2(got['branches'][mi][k], ref) for k, ref in g['branches'][mi].items()})
S = core.encoders[2].bf
for key, t in S.bufs.items():
    if t.dtype == torch.int16:
        v = bf16.to_numpy(t)
        print(key, tuple(t.shape), "nan" if not np.isfinite(v).all() else "ok", "%.3g" % np.abs(v[np.isfinite(v)]).max())
    elif t.dtype == torch.float32:
        v = t.cpu().numpy()
        print(key, tuple(t.shape), "f32", "nan" if not np.isfinite(v).all() else "ok", "%.3g" % np.abs(v[np.isfinite(v)]).max())
