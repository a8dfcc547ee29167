"""Builds libugaitnet_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m ugaitnet_amd.build [--force] [--h2]

--h2 (or UGN_BUILD_H2=1): also build the opt-in f16x2 ("H2") kernel set of rounds 3-4 (conv3x3_mm.hip, wgrad3x3_mm.hip,
h2_elem.hip + the h2 entry points of conv5x5.hip / pool_set.hip; include/ugaitnet_hip_h2.h).  The default library does not carry it:
it is narrower than the reference's fp32, batch-dependent and credited nowhere, and it was the longest part of the build.
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
# (longest compiles first: four run at a time)
# conv3x3.hip (the direct fp32-MFMA set: 48 data-gradient instantiations, 150 s as one object) is compiled as five objects:
# "file@k" = that file with -DUGN_C3_PART=k into <file>_p<k>.o
SOURCES = ["conv3x3.hip@1", "conv3x3.hip@2", "conv3x3.hip@3", "conv3x3.hip@4", "conv3x3_x3.hip", "conv3x3.hip@0", "conv3x3_wino.hip", "wgrad3x3_x3.hip",
           "conv5x5.hip", "conv3x3_bf.hip", "wgrad3x3_bf.hip", "wgrad3x3_wino.hip", "conv3x3_wino_tall.hip", "head.hip", "pool_set.hip",
           "bf_elem.hip", "knn.hip", "assemble.hip", "error.cpp", "runtime.cpp"]
H2_SOURCES = ["conv3x3_mm.hip", "wgrad3x3_mm.hip", "h2_elem.hip"]       # the opt-in f16x2 set (--h2 / UGN_BUILD_H2=1)
H2_FLAGGED = ("conv5x5.hip", "pool_set.hip")                           # sources that carry `#if UGN_WITH_H2` entry points
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "wino_common.h"), os.path.join(CSRC, "mm_common.h"), os.path.join(CSRC, "x3_common.h"), os.path.join(HERE, "..", "include", "ugaitnet_hip.h"), os.path.join(HERE, "..", "include", "ugaitnet_hip_h2.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
FLAGS += os.environ.get("UGN_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DUGN_...); the default build sets none


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, flags):
    """sha256 over the CONTENT of a source, every header it may include and the compiler command: an object is reused only when its
    recorded digest matches (modification times say nothing after a checkout, a copy or a snapshot to another machine)."""
    import hashlib
    h = hashlib.sha256()
    h.update("\0".join(flags).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(b"\0" + os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def _compile(src, objdir=None, extra=(), suffix=""):
    import glob
    obj = os.path.join(objdir or OBJ, os.path.splitext(src)[0] + suffix + ".o")
    path = os.path.join(CSRC, src)
    cmd = [HIPCC] + FLAGS + list(extra) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    headers = sorted(set(HEADERS + glob.glob(os.path.join(CSRC, "*.h"))))
    want = _digest([path] + headers, cmd[:-3])
    stamp = obj + ".sha256"
    have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(obj) else ""
    if have != want:
        import time
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if os.environ.get("UGN_BUILD_TIMES"):
            print("  %-24s %.1f s" % (src, time.perf_counter() - t0), flush=True)
        with open(stamp, "w") as f:
            f.write(want + "\n")
        return obj, True
    return obj, False


def with_h2():
    return os.environ.get("UGN_BUILD_H2", "0") not in ("", "0")


def build(force=False, verbose=True, variant=None, extra=(), h2=None):
    """h2: also build the opt-in f16x2 set (None: what UGN_BUILD_H2 says; default off).
    variant (experiments only): build libugaitnet_hip_<variant>.so with `extra` compiler flags into its own object directory;
    ugaitnet_amd._lib loads it when UGN_LIB names it (tools/ab_ops.py times two builds side by side on one GPU box)."""
    objdir = OBJ if variant is None else OBJ + "_" + variant
    LIB = globals()["LIB"] if variant is None else os.path.join(HERE, "libugaitnet_hip_%s.so" % variant)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)      # (a variant links the default build's objects for sources its macros do not touch)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
        if os.path.exists(LIB + ".sha256"):
            os.remove(LIB + ".sha256")
    # a variant recompiles only the sources (or headers) that mention one of its -D macros; the rest links the default build's objects
    import glob
    import re
    macros = [f[2:].split("=")[0] for f in extra if f.startswith("-D")]
    word = lambda m, text: re.search(r"\b%s\b" % re.escape(m), text) is not None
    all_headers = sorted(set(HEADERS + glob.glob(os.path.join(CSRC, "*.h"))))
    hdr_hit = any(word(m, open(h).read()) for h in all_headers for m in macros)

    h2 = with_h2() if h2 is None else bool(h2)
    sources = (H2_SOURCES + SOURCES) if h2 else SOURCES

    def one(src):
        if "@" in src:                        # one part of a source compiled in several objects
            name, part = src.split("@")
            return _compile(name, objdir if variant is not None else OBJ, tuple(extra) + ("-DUGN_C3_PART=" + part,), suffix="_p" + part)
        if h2 and src in H2_FLAGGED:         # (an object of its own name: toggling the option recompiles nothing)
            return _compile(src, objdir, tuple(extra) + ("-DUGN_WITH_H2=1",), suffix="_h2")
        if variant is not None and macros and len(macros) == len(list(extra)) and not hdr_hit:
            if not any(word(m, open(os.path.join(CSRC, src)).read()) for m in macros):
                return _compile(src, OBJ, ())
        return _compile(src, objdir, extra)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("UGN_BUILD_JOBS", "6"))) as ex:
        res = list(ex.map(one, sources))
    objs = [o for o, _ in res]
    # the library records the digests of the objects it was linked from: an unchanged tree links nothing, anything else re-links
    link_want = "\n".join(open(o + ".sha256").read().strip() for o in objs)
    link_stamp = LIB + ".sha256"
    link_have = open(link_stamp).read().strip() if os.path.exists(link_stamp) and os.path.exists(LIB) else ""
    if any(c for _, c in res) or link_have != link_want:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        with open(link_stamp, "w") as f:
            f.write(link_want + "\n")
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a not in ("--force", "--h2")]
    h2 = True if "--h2" in sys.argv else None
    if args and args[0] == "--variant":      # python -m ugaitnet_amd.build --variant NAME -DFLAG ...
        build(force="--force" in sys.argv, variant=args[1], extra=args[2:], h2=h2)
    else:
        build(force="--force" in sys.argv, h2=h2)
