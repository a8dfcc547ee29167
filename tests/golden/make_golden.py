"""Generates tests/golden/*.npz from the CPU oracle (fp64 master run).

PARITY UNPINNED: the reference cannot be imported here (TensorFlow 2.3 is not installed) and it
ships no fixtures, so these vectors come from oracle/ugaitnet_oracle.py, whose semantics are pinned by the independent
torch-autograd implementation and the known-answer tests.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ugaitnet_oracle as O  # noqa: E402
from tests.synth import make_batch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def params_for(kinds, ncls, seed):
    rng = np.random.default_rng(seed)
    p = dict(branches=[O.init_branch_params(rng, 2 if k == 'of' else 1, np.float64) for k in kinds])
    if ncls:
        p['head'] = O.init_head_params(rng, ncls, np.float64)
        p['head']['bc'] = rng.normal(size=ncls) * 0.01
    return p


def digest(params):
    h = hashlib.sha256()
    for bp in params['branches']:
        for k in sorted(bp):
            h.update(bp[k].astype(np.float32).tobytes())
    if 'head' in params:
        for k in sorted(params['head']):
            h.update(params['head'][k].astype(np.float32).tobytes())
    return h.hexdigest()


def case(name, kinds, b, l, ncls, ids, mode, multimodal, seed, masks=True):
    xs, uses, labels, onehot = make_batch(kinds, b, l, max(ncls, 1), ids=ids, seed=seed, masks=masks)
    p = params_for(kinds, ncls, seed + 1)
    r, g = O.model_loss_and_grads([x.astype(np.float64) for x in xs], [u.astype(np.float64) for u in uses] if multimodal else None,
                                  labels, onehot.astype(np.float64), p, margin=0.2, loss_weights=(1.0, 0.1), mode=mode,
                                  multimodal=multimodal)
    out = dict(kinds=np.array(kinds), b=b, l=l, ncls=ncls, ids=ids, mode=mode, multimodal=multimodal, seed=seed,
               param_seed=seed + 1, param_sha256=digest(p), labels=labels, onehot=onehot,
               signature=r['signature'].astype(np.float32), loss=np.float64(r['loss']), triplet=np.float64(r['triplet']),
               active_triplets=r['tri_aux']['num'].astype(np.float32), hp=r['tri_aux']['hp'], hn=r['tri_aux']['hn'],
               kp=r['tri_aux']['kp'], kn=r['tri_aux']['kn'])
    for i, (x, u) in enumerate(zip(xs, uses)):
        out['x%d' % i] = x
        out['use%d' % i] = u
    if ncls:
        out.update(logits=r['logits'].astype(np.float32), probs=r['probs'].astype(np.float32), xent=np.float64(r['xent']))
        out['grad_head_bc'] = g['head']['bc']
    if multimodal and mode != 'avg':
        out['sel'] = r['sel'].astype(np.uint8)
    for i, gb in enumerate(g['branches']):
        out['grad_m%d_a1' % i] = gb['a1']                       # small tensors kept whole
        out['grad_m%d_a6_l2' % i] = np.linalg.norm(gb['a6'])    # large ones as norms
        out['grad_m%d_fc_l2' % i] = np.linalg.norm(gb['fc'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'loss', float(r['loss']), 'params', out['param_sha256'][:12])


if __name__ == '__main__':
    case('c3_3mod_signmax', ('of', 'gray', 'depth'), 4, 2, 6, 2, 'sign_max', True, 11)
    case('c1_1mod_gray', ('gray',), 4, 2, 6, 2, 'sign_max', False, 12)
    case('c4_3mod_sil_signmax', ('of', 'gray', 'sil'), 4, 2, 5, 2, 'sign_max', True, 13)
    # keras Maximum fusion, no missing modality (with masked rows the batch-axis norm of a column can be ~0, which
    # amplifies fp32 rounding beyond the 1e-3 band at this tiny batch size)
    case('c2mod_of_gray_max', ('of', 'gray'), 4, 2, 5, 2, 'max', True, 14, masks=False)
