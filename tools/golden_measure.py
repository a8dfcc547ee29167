import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import ugaitnet_oracle as O
from tests.test_golden import load, FILES
from ugaitnet_amd import ops
from ugaitnet_amd.engine import GaitCore
for path in FILES:
    z, kinds, xs, uses, p = load(path)
    mm = bool(z['multimodal'])
    core = GaitCore([x.shape[-1] for x in xs], nclasses=int(z['ncls']), multimodal=mm, fuse_mode=str(z['mode']), margin=0.2, loss_weights=(1.0, 0.1))
    core.set_params_numpy(O.cast_params(p, np.float32))
    core.forward_backward(xs, uses if mm else None, z['labels'], z['onehot'])
    ls = core.losses(); got = core.get_grads_numpy()
    out = [os.path.basename(path), 'sig %.2e' % np.abs(core.sig.cpu().numpy() - z['signature']).max(), 'probs %.2e' % np.abs(core.head['probs'].cpu().numpy() - z['probs']).max(),
           'loss %.2e' % abs(ls['loss'] - float(z['loss'])), 'tri %.2e' % abs(ls['triplet'] - float(z['triplet']))]
    for i in range(len(kinds)):
        ref = z['grad_m%d_a1' % i]
        out.append('a1[%d] %.2e' % (i, np.linalg.norm(got['branches'][i]['a1'] - ref) / np.linalg.norm(ref)))
        out.append('fc[%d] %.2e' % (i, abs(np.linalg.norm(got['branches'][i]['fc']) - float(z['grad_m%d_fc_l2' % i])) / float(z['grad_m%d_fc_l2' % i])))
    if 'sel' in z.files:
        out.append('sel mismatch %d of %d' % ((core.sel.cpu().numpy() != z['sel']).sum(), z['sel'].size))
    print(' | '.join(out), flush=True)
