"""Keras-HDF5 checkpoints for the gaitset graph, on top of ugaitnet_amd/h5lite.py.

The reference trains with `ModelCheckpoint(save_weights_only=True)`, `model.save_weights(...hdf5)` and restores with
`model.load_weights(filewes, by_name=True, skip_mismatch=True)` (mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:364,524-538;
nets/mj_uwyhNets_ba.py:630,644,1027).  A Keras weights file lists the layers in the root attribute `layer_names`; each layer
is a group with a `weight_names` attribute and its arrays below `<layer>/<weight name>`; `model.save` keeps the same tree
below `/model_weights`.

The gaitset branch names none of its layers (nets/mj_uwyhNets_ba.py:428-482), so the file carries Keras' automatic names.
Those count per class in creation order: modality m's branch creates TimeDistributed wrappers 15m .. 15m+14 and Conv2D layers
10m .. 10m+9; the wrapped convolutions store their kernel under the wrapper's name (`time_distributed_1/kernel:0`), the
set-level ones under `conv2d_2` ...; `MatMul` is `mat_mul`, `mat_mul_1`, ... with one variable of a random name (:31); the
classifier is the layer named `classprob` (:850).  A model built after other models in the same process has shifted
counters, so files are matched per class by ascending counter -- which equals the name when the counters start at 0.
"""
from __future__ import annotations

import re

import numpy as np

from . import h5lite

TD_SLOT = {"a1": 1, "a2": 3, "a3": 6, "a4": 8, "a5": 11, "a6": 13}    # index of the wrapper among the branch's 15
CONV_SLOT = {"b1": 2, "b2": 3, "b3": 6, "b4": 7}                      # index of the layer among the branch's 10 Conv2D
TD_ORDER = ("a1", "a2", "a3", "a4", "a5", "a6")
CONV_ORDER = ("b1", "b2", "b3", "b4")
TD_PER_BRANCH, CONV_PER_BRANCH = 15, 10


def _auto(base, idx):
    return base if idx == 0 else "%s_%d" % (base, idx)


def keras_layer_name(mi, pname):
    """Keras' automatic name of parameter `pname` (a1..a6, b1..b4, fc) of modality `mi` in a freshly started process."""
    if pname in TD_SLOT:
        return _auto("time_distributed", TD_PER_BRANCH * mi + TD_SLOT[pname])
    if pname in CONV_SLOT:
        return _auto("conv2d", CONV_PER_BRANCH * mi + CONV_SLOT[pname])
    if pname == "fc":
        return _auto("mat_mul", mi)
    raise KeyError(pname)


def read_layers(path):
    """[(layer name, [(weight name, array)])] in the file's `layer_names` order (root or /model_weights)."""
    f = h5lite.File(path)
    g = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
    if "layer_names" not in g.attrs:
        raise ValueError("%s: no layer_names attribute -- not a Keras weights file" % path)
    as_str = lambda v: v.decode("utf-8") if isinstance(v, bytes) else str(v)
    out = []
    for lname in [as_str(n) for n in np.atleast_1d(g.attrs["layer_names"])]:
        lg = g[lname]
        wnames = [as_str(n) for n in np.atleast_1d(lg.attrs.get("weight_names", []))]
        out.append((lname, [(w, lg[w].read()) for w in wnames]))
    return out


def _counter(name, base):
    m = re.fullmatch(re.escape(base) + r"(?:_(\d+))?", name)
    return None if m is None else int(m.group(1) or 0)


def assign(layers, nmod, nclasses):
    """Map a Keras file's layers onto this build's parameter names.  Returns {param name: array}: 'm<i>.<a1..fc>', 'head.wc',
    'head.bc'.  Layers are taken per class in ascending counter order (see module docstring)."""
    by_class = {"time_distributed": [], "conv2d": [], "mat_mul": []}
    out = {}
    for lname, ws in layers:
        if not ws:
            continue
        if lname == "classprob":
            if nclasses > 0:
                for w, a in ws:
                    out["head.wc" if a.ndim == 2 else "head.bc"] = a
            continue
        for base in by_class:
            c = _counter(lname, base)
            if c is not None:
                by_class[base].append((c, ws[0][1]))
    for base, order in (("time_distributed", TD_ORDER), ("conv2d", CONV_ORDER), ("mat_mul", ("fc",))):
        arrs = [a for _, a in sorted(by_class[base], key=lambda t: t[0])]
        for mi in range(nmod):
            for j, pname in enumerate(order):
                k = mi * len(order) + j
                if k < len(arrs):
                    out["m%d.%s" % (mi, pname)] = arrs[k]
    return out


def write_weights(path, params, in_channels, nclasses, extra=None, below=""):
    """Write {param name: array} in the Keras layout, with the names a freshly built reference model would carry (so that its
    `load_weights(by_name=True)` finds them).  `extra`: {path: array} written as plain datasets beside the weights; `below`:
    'model_weights' for the layout of `model.save`."""
    w = h5lite.Writer()
    root = below.strip("/")
    pre = root + "/" if root else ""
    w.create_group(root)
    names = []
    for mi in range(len(in_channels)):
        for pname in ("a1", "a2", "b1", "b2", "a3", "a4", "b3", "b4", "a5", "a6", "fc"):
            lname = keras_layer_name(mi, pname)
            wname = "MatMul_kernel[%d]:0" % mi if pname == "fc" else lname + "/kernel:0"
            names.append(lname)
            w.create_dataset(pre + lname + "/" + wname, np.asarray(params["m%d.%s" % (mi, pname)], np.float32))
            w.set_attr(pre + lname, "weight_names", [wname.encode()])
    if nclasses > 0:
        names.append("classprob")
        w.create_dataset(pre + "classprob/classprob/kernel:0", np.asarray(params["head.wc"], np.float32))
        w.create_dataset(pre + "classprob/classprob/bias:0", np.asarray(params["head.bc"], np.float32))
        w.set_attr(pre + "classprob", "weight_names", [b"classprob/kernel:0", b"classprob/bias:0"])
    w.set_attr(root, "layer_names", [n.encode() for n in names])
    w.set_attr(root, "backend", b"tensorflow")
    w.set_attr(root, "keras_version", b"2.4.0")
    for k, v in (extra or {}).items():
        if isinstance(v, (str, bytes)):
            w.set_attr("", k, v)
        else:
            w.create_dataset(k, v)
    w.save(path)
