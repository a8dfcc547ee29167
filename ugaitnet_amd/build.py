"""Builds libugaitnet_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m ugaitnet_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libugaitnet_hip.so")
SOURCES = ["conv3x3.hip", "conv3x3_wino.hip", "conv3x3_wino_tall.hip", "wgrad3x3_wino.hip", "conv5x5.hip", "pool_set.hip", "head.hip", "knn.hip", "assemble.hip", "error.cpp"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "wino_common.h"), os.path.join(HERE, "..", "include", "ugaitnet_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
FLAGS += os.environ.get("UGN_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DUGN_...); the default build sets none


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
