"""Synthetic batches shaped like the reference's generators, shared by tests and bench.py.

Batch contract (data/mj_dataGeneratorMMUWYHsingle_repetitions.py:331-345,530-545,658-675): X = [mod_i, use_i]*,
y = [labels [B,1], onehot [B,ncls]]; a disabled modality is the constant 1e-9 with flag 0 (:102,414-415).
"""
import numpy as np

MASK_PATTERNS = ((1, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1))


def make_batch(kinds, b, l=25, nclasses=150, ids=None, per_id=2, seed=232323, masks=True, dtype=np.float32, all_present=False):
    """kinds: tuple of 'of' | 'gray' | 'depth' | 'sil'.  Returns (xs, uses, labels, onehot).
    all_present: no modality dropping (every `use` flag 1, no 1e-9 placeholder tensors): bench.py's value_all_present."""
    masks = masks and not all_present
    rng = np.random.default_rng(seed)
    xs = []
    for k in kinds:
        if k == 'of':      # int16 optical flow / compressFactor 100 * 0.1
            x = np.round(rng.normal(0, 300, (b, l, 60, 60, 2))).astype(np.int16).astype(dtype) * dtype(0.001)
        elif k == 'sil':   # binary silhouettes
            x = (rng.uniform(size=(b, l, 60, 60, 1)) < 0.3).astype(dtype)
        else:              # gray / depth: uint8/255 - 0.5
            x = rng.uniform(-0.5, 0.5, (b, l, 60, 60, 1)).astype(dtype)
        xs.append(x)
    nmod = len(kinds)
    uses = [np.ones((b, 1), dtype) for _ in kinds]
    if masks and nmod > 1:
        for r in range(b):
            pat = MASK_PATTERNS[r % 7] if nmod == 3 else ((1, 1), (1, 0), (0, 1))[r % 3]
            for m in range(nmod):
                if not pat[m]:
                    uses[m][r, 0] = 0
                    xs[m][r] = 1e-9
    ids = b // per_id if ids is None else ids
    labels = np.repeat(np.arange(ids), b // ids)[:b].astype(np.int64)
    onehot = np.eye(nclasses, dtype=dtype)[labels]
    return xs, uses, labels, onehot
