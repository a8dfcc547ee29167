"""Eager launches vs the captured-graph step (engine.GraphedTrainStep) at several per-GPU batch sizes, C3 shape.
usage: bench_graph.py [f32|bf16] [clips ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.synth import make_batch
from ugaitnet_amd.engine import GaitCore, GraphedTrainStep

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
sizes = [int(a) for a in sys.argv[2:]] or [4, 8, 16, 24]
for b in sizes:
    xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), b, 25, 150, seed=232323)
    dxs = [torch.from_numpy(x).cuda() for x in xs]
    dus = [torch.from_numpy(u).cuda() for u in uses]
    doh = torch.from_numpy(onehot).cuda()
    row = []
    for mode in ("eager", "graph"):
        core = GaitCore([2, 1, 1], nclasses=150, fuse_mode="sign_max", margin=0.2, loss_weights=(1.0, 0.1), seed=1, conv_precision=prec)
        step = GraphedTrainStep(core, dxs, dus, labels, doh).step if mode == "graph" else core.train_step
        for _ in range(5):
            step(dxs, dus, labels, doh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            step(dxs, dus, labels, doh)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / n * 1e3)
        del core
    print("%s clips %3d: eager %.3f ms (%.0f clips/s)  graph %.3f ms (%.0f clips/s)" % (prec, b, row[0], b / row[0] * 1e3, row[1], b / row[1] * 1e3), flush=True)
