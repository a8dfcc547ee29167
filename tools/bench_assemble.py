"""Device-side batch assembly: time of ugn_assemble_modality for one batch, its HBM rate and the H2D
sizes of the raw and the assembled routes."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ugaitnet_amd.batching import DeviceBatchAssembler, ModalitySpec, plan_rows

nbase, expand = 8, 3
rng = np.random.default_rng(0)
samples = [[rng.integers(-3000, 3000, (60, 60, 50)).astype(np.int16), rng.integers(0, 256, (60, 60, 25)).astype(np.uint8),
            rng.integers(0, 256, (60, 60, 25)).astype(np.uint8)] for _ in range(nbase)]
present = np.ones((nbase, 3), bool)
plan = plan_rows(present, expand, rng=random.Random(1))
raws = [torch.from_numpy(np.stack([s[j] for s in samples])).cuda() for j in range(3)]
asm = DeviceBatchAssembler([ModalitySpec("of", 2, compress_factor=100.0), ModalitySpec("gray", 1), ModalitySpec("depth", 1)])
asm.assemble(raws, plan)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = asm.assemble(raws, plan)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
wr = sum(x.numel() * 4 for x, _ in out)
rd = sum(int((plan[:, j] >= 0).sum()) * raws[j][0].numel() * raws[j].element_size() for j in range(3))
print("GPU assembly of %d rows x 3 modalities: %.1f us (incl. 3 launches + plan upload), %.1f MB written + %.1f MB read = %.2f TB/s" % (
    plan.shape[0], us, wr / 1e6, rd / 1e6, (wr + rd) / us / 1e6))
print("H2D per batch: raw %.1f MB vs assembled float32 %.1f MB (the reference ships float64: %.1f MB)" % (
    sum(r.numel() * r.element_size() for r in raws) / 1e6, wr / 1e6, 2 * wr / 1e6))
