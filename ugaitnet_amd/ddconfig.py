"""`model-config.hdf5`: the architecture dictionary the reference's mains save with `deepdish.io.save`
(mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:474-489) and its nets read back with `deepdish.io.load` for the
"surgery" route (nets/mj_uwyhNets_ba.py:555-579, :610-630).  deepdish / PyTables are not installed here (and cannot be), so
the file layout is restated from deepdish's `hdf5io` module (the 0.3 series the reference's Python 3 / TF 2.3 era used) and read
and written with `h5lite`:

* a dictionary with string keys is stored "flat" in the root group: every entry under its key;
* int / float / bool / str / bytes and numpy scalars become ATTRIBUTES of the group that holds them (PyTables stores them as
  scalar HDF5 attributes: int64, float64, an int8 enumeration FALSE/TRUE, a fixed-length string);
* a list / tuple / dict becomes a GROUP whose `TITLE` attribute is `list:N` / `tuple:N` / `dict:N`; the elements of a
  sequence are named `i0` ... `i{N-1}` and stored by the same rules (scalars as attributes of that group, the rest as nodes);
* `None` is an empty group titled `nonetype:`; a numpy array is an array node; an EMPTY array is the array of its shape with
  the node attribute `zeroarray_dtype`;
* anything else deepdish pickles into a node (`pyobject:`) -- refused here, nothing in the model configuration needs it;
* the root carries `DEEPDISH_IO_VERSION` (12); a non-dictionary top level would be stored as `data` + `DEEPDISH_IO_UNPACK`.

PARITY UNPINNED: no file written by deepdish exists in the reference tree or in this image to check the reader against; the
writer and the reader are checked against each other and against the layout above (tests/test_ddconfig.py).  A value that
does not parse raises `h5lite.H5Error` / `ValueError` naming the key instead of guessing.
"""
from __future__ import annotations

import numpy as np

from . import h5lite

IO_VERSION = 12
_BOOKKEEPING = ("CLASS", "VERSION", "TITLE", "FLAVOR", "PYTABLES_FORMAT_VERSION", "FILTERS", "DEEPDISH_IO_VERSION",
                "DEEPDISH_IO_UNPACK", "DEEPDISH_IO_ROOT_IS_SNS")


# ---------------------------------------------------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------------------------------------------------
def _scalar(v, where):
    """An attribute value as the Python object deepdish hands back."""
    if isinstance(v, (bytes, np.bytes_)):
        b = bytes(v)
        if b[:1] == b"\x80":      # PyTables pickles what it cannot store natively; a configuration holds nothing of the kind
            raise ValueError("%s: pickled attribute (a Python object deepdish could not store natively)" % where)
        return b.decode("utf-8")
    if isinstance(v, str):
        return v
    a = np.asarray(v)
    if a.ndim == 0:
        return a.item()
    return a


def _title(node):
    t = node.attrs.get("TITLE", b"")
    return t.decode("utf-8") if isinstance(t, (bytes, np.bytes_)) else str(t)


def _load_node(node, where):
    if isinstance(node, h5lite.Dataset):
        a = node.read()
        zdt = node.attrs.get("zeroarray_dtype")
        if zdt is not None:
            zdt = zdt.decode("ascii") if isinstance(zdt, (bytes, np.bytes_)) else str(zdt)
            return np.zeros(tuple(int(v) for v in np.asarray(a).reshape(-1)), dtype=np.dtype(zdt))
        return a
    title = _title(node)
    kind, _, count = title.partition(":")
    if kind == "nonetype":
        return None
    if kind in ("list", "tuple"):
        n = int(count)
        out = [_load_entry(node, "i%d" % i, "%s/i%d" % (where, i)) for i in range(n)]
        return out if kind == "list" else tuple(out)
    if kind in ("dict", ""):
        return _load_dict(node, where)
    if kind in ("pyobject", "sns", "sparse"):
        raise ValueError("%s: a %s node (pickled / namespace / sparse object) is not part of a model configuration" % (where, kind))
    raise ValueError("%s: unknown deepdish node title %r" % (where, title))


def _load_entry(group, name, where):
    if name in group.attrs:
        return _scalar(group.attrs[name], where)
    if name in group:
        return _load_node(group[name], where)
    raise ValueError("%s: missing" % where)


def _load_dict(group, where):
    out = {}
    for k, v in group.attrs.items():
        if k not in _BOOKKEEPING:
            out[k] = _scalar(v, "%s/%s" % (where, k))
    for k in group.keys():
        out[k] = _load_node(group[k], "%s/%s" % (where, k))
    return out


def load(path):
    """What `deepdish.io.load(path)` returns for a file saved from a dictionary (or any natively storable value)."""
    f = h5lite.File(path)
    d = _load_dict(f, "")
    unpack = f.attrs.get("DEEPDISH_IO_UNPACK")
    if unpack is not None and bool(np.asarray(unpack).reshape(-1)[0]) and "data" in d:
        return d["data"]
    return d


# ---------------------------------------------------------------------------------------------------------------------
# writing
# ---------------------------------------------------------------------------------------------------------------------
_SCALARS = (bool, int, float, str, bytes, np.integer, np.floating, np.bool_)


def _attr_value(v):
    if isinstance(v, (bool, np.bool_)):
        return h5lite.Bool(v)              # the int8 enumeration FALSE / TRUE, as PyTables stores numpy.bool_
    if isinstance(v, (int, np.integer)):
        return np.int64(v)
    if isinstance(v, (float, np.floating)):
        return np.float64(v)
    return v


def _save_entry(w, group_path, name, v):
    path = (group_path + "/" + name) if group_path else name
    if isinstance(v, _SCALARS):
        w.set_attr(group_path, name, _attr_value(v))
    elif v is None:
        w.create_group(path)
        w.set_attr(path, "TITLE", "nonetype:")
    elif isinstance(v, (list, tuple)):
        w.create_group(path)
        w.set_attr(path, "TITLE", "%s:%d" % ("list" if isinstance(v, list) else "tuple", len(v)))
        for i, e in enumerate(v):
            _save_entry(w, path, "i%d" % i, e)
    elif isinstance(v, dict):
        if not all(isinstance(k, str) for k in v):
            raise ValueError("%s: dictionary keys must be strings" % path)
        w.create_group(path)
        w.set_attr(path, "TITLE", "dict:%d" % len(v))
        for k, e in v.items():
            _save_entry(w, path, k, e)
    elif isinstance(v, np.ndarray):
        if v.size == 0:
            w.create_dataset(path, np.asarray(v.shape, dtype=np.int64))
            w.set_attr(path, "zeroarray_dtype", v.dtype.str)
        else:
            w.create_dataset(path, v)
    else:
        raise ValueError("%s: a %s cannot be stored natively (deepdish would pickle it); pass its name instead"
                         % (path, type(v).__name__))


def save(path, data):
    """`deepdish.io.save(path, data)` for a dictionary with string keys (the model configuration)."""
    w = h5lite.Writer()
    w.set_attr("", "DEEPDISH_IO_VERSION", np.int64(IO_VERSION))
    w.set_attr("", "TITLE", "")
    if isinstance(data, dict) and all(isinstance(k, str) for k in data):
        for k, v in data.items():
            _save_entry(w, "", k, v)
    else:
        _save_entry(w, "", "data", data)
        w.set_attr("", "DEEPDISH_IO_UNPACK", h5lite.Bool(True))
    w.save(path)
