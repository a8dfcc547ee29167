"""Diagnostic: per modality and pooled layer, how the HIP path's MaxPool argmax compares with the fp64 oracle on the tie-heavy
batches of tests/test_ties_gpu.py (exactly tied windows routed differently; mismatches elsewhere; value errors)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import test_ties_gpu as T


def main():
    import torch
    dev = torch.device("cuda:0")
    for name, batch in (("flat", T._flat_batch()), ("diag", T._diag_batch())):
        for direct in (False, True):
            core, r, g = T._run(dev, batch, direct=direct)
            print("== %s batch, %s kernels" % (name, "direct" if direct else "Winograd"))
            for mi, enc in enumerate(core.encoders):
                c = r["branch"][mi]
                for pre, ref_idx, key, val in ((c["a2"], c["i2"], "i2", "p2"), (c["a4"], c["i4"], "i4", "p4"), (c["b2"], c["j2"], "j2", "q2")):
                    got = enc.act[key].cpu().numpy()
                    ties, first = T._tie_windows(pre)
                    verr = np.abs(enc.act[val].cpu().numpy() - c[val]).max() / (np.abs(c[val]).max() + 1e-300)
                    # near-ties: windows whose two largest fp64 values differ by less than 1e-6 of the map's scale
                    n, h, w, ch = pre.shape
                    xw = np.sort(pre.reshape(n, h // 2, 2, w // 2, 2, ch).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // 2, w // 2, 4, ch), axis=3)
                    gap = (xw[:, :, :, 3] - xw[:, :, :, 2]) / (np.abs(pre).max() + 1e-300)
                    other = (got != ref_idx) & ~ties
                    print("  mod %d %s: windows %d, exact ties %d (moved %d), other mismatches %d (of which gap < 1e-6: %d, max gap %.2e), "
                          "pooled value rel err %.2e" % (mi, key, ties.size, ties.sum(), (got[ties] != first[ties]).sum(), other.sum(),
                                                         (other & (gap < 1e-6)).sum(), gap[other].max() if other.any() else 0.0, verr))
            worst = T._grad_errors(core, g)
            print("  worst gradient rel-L2 %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))


if __name__ == "__main__":
    main()
