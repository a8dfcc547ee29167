#!/usr/bin/env python3
"""Time single operators of two (or more) builds of libugaitnet_hip.so side by side on one GPU box.

    python tools/ab_ops.py OP[,OP...] LIB_A.so LIB_B.so [--frames 600] [--reps 30] [--rounds 3]

Each (library, round) runs in its own process (UGN_LIB selects the build), rounds interleaved A B A B so that clock and thermal
drift hit both alike; per op the median of the per-launch HIP-event times of every round is printed.
Ops: conv5x5_fwd1 conv5x5_fwd2 conv5x5_wgrad1 conv5x5_wgrad2 (conv5x5x_*: the same in the x3 arithmetic) a2_fwd a2_dgrad a2_wgrad a3_fwd a3_dgrad a3_wgrad a4_fwd a4_dgrad
a4_wgrad a5_* a6_* setmax_fwd setmax_bwd step (whole C3 training step, 24 clips)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = {"a2": (64, 32, 32, True), "a3": (32, 32, 64, False), "a4": (32, 64, 64, True), "a5": (16, 64, 128, False), "a6": (16, 128, 128, False)}


def child(ops_list, frames, reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from ugaitnet_amd import ops
    dev = torch.device("cuda")
    out = {}

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ts]) * 1e3)

    for op in ops_list:
        if op.startswith("conv5x5"):
            cin = int(op[-1])
            x3 = op.startswith("conv5x5x")       # conv5x5x_fwd1 ...: the x3 arithmetic form (ugn_x3_conv5x5_in_*)
            x = torch.rand(frames, 60, 60, cin, device=dev) - 0.5
            w = torch.randn(5, 5, cin, 32, device=dev) * 0.1
            a1 = torch.empty(frames, 64, 64, 32, device=dev)
            sg = torch.empty(frames, 64, 64, dtype=torch.int32, device=dev)
            if "fwd" in op:
                out[op] = timeit(lambda: ops.conv5x5_in_fwd(x, w, a1, sign=sg, x3=x3))
            else:
                ops.conv5x5_in_fwd(x, w, a1, sign=sg)
                dz = torch.randn(frames, 64, 64, 32, device=dev)
                dw = torch.empty(5, 5, cin, 32, device=dev)
                out[op] = timeit(lambda: ops.conv5x5_in_wgrad(x, dz, dw, sign=sg, x3=x3))
        elif op == "step":
            from tests.synth import make_batch
            from ugaitnet_amd.engine import GaitCore
            xs, uses, labels, onehot = make_batch(("of", "gray", "depth"), 24, 25, 150, seed=232323)
            core = GaitCore([2, 1, 1], nclasses=150, loss_weights=(1.0, 0.1), seed=1)
            dxs = [torch.from_numpy(x).to(dev) for x in xs]
            dus = [torch.from_numpy(u).to(dev) for u in uses]
            doh = torch.from_numpy(onehot).to(dev)
            out[op] = timeit(lambda: core.train_step(dxs, dus, labels, doh))
        elif op.startswith("setmax"):
            b, l, s = 24, 25, 32768
            p = torch.randn(b * l, s, device=dev)
            m = torch.empty(b, s, device=dev)
            if op == "setmax_fwd":
                out[op] = timeit(lambda: ops.setmax_fwd(p, b, l, m=m))
            else:
                dm = torch.randn(b, s, device=dev)
                o = torch.empty_like(p)
                out[op] = timeit(lambda: ops.setmax_bwd(p, dm, b, l, True, out=o, addend=o))
        else:
            layer, kind = op.split("_")
            hw, cin, cout, pool = CFGS[layer]
            n = frames
            x = torch.randn(n, hw, hw, cin, device=dev)
            w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
            ho = hw // 2 if pool else hw
            dz = torch.randn(n, ho, ho, cout, device=dev)
            idx = torch.randint(0, 4, (n, ho, ho, cout), device=dev, dtype=torch.uint8) if pool else None
            if kind == "fwd":
                uf = ops.wino_pack(w, False)
                o = torch.empty(n, ho, ho, cout, device=dev)
                oi = torch.empty(n, ho, ho, cout, device=dev, dtype=torch.uint8) if pool else None
                out[op] = timeit(lambda: ops.conv3x3_fwd_wino(x, uf, cout, pool, o, oi))
            elif kind == "dgrad":
                ud = ops.wino_pack(w, True, pooled_dz=pool)
                o = torch.empty(n, hw, hw, cin, device=dev)
                act = torch.randn(n, hw, hw, cin, device=dev) if layer in ("a4", "a6") else None   # (the engine runs a2, a3, a5 with the plain epilogue)
                out[op] = timeit(lambda: ops.conv3x3_dgrad_wino(dz, ud, hw, cin, cout, dz_idx=idx, act=act, out=o))
            else:
                dw = torch.empty(3, 3, cin, cout, device=dev)
                out[op] = timeit(lambda: ops.conv3x3_wgrad_wino(x, dz, cout, dz_idx=idx, dw=dw))
    print("AB_RESULT " + json.dumps(out))


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        child(argv[1].split(","), int(argv[2]), int(argv[3]))
        return
    opt = lambda k, d: int(argv[argv.index(k) + 1]) if k in argv else d
    frames, reps, rounds = opt("--frames", 600), opt("--reps", 30), opt("--rounds", 3)
    pos = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or not argv[i - 1].startswith("--"))]
    ops_arg, libs = pos[0], pos[1:]
    res = {lib: [] for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            env = dict(os.environ, UGN_LIB=os.path.abspath(lib))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ops_arg, str(frames), str(reps)], env=env,
                               capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if r.returncode != 0 or not line:
                print(lib, "FAILED", r.stderr[-1500:])
                continue
            res[lib].append(json.loads(line[0][10:]))
    for op in ops_arg.split(","):
        print("%-16s" % op + "  ".join("%s: %s us" % (os.path.basename(lib), "/".join("%.1f" % r[op] for r in res[lib] if op in r)) for lib in libs))


if __name__ == "__main__":
    main()
