"""Reader for the reference's on-disk samples (file format only).

`data/generateOFData.py:137-149` / `data/generateRGBData.py:171-177` write one deepdish (PyTables-flavoured HDF5) file per
sample: arrays become nodes (`data` int16 [60,60,50] for optical flow, uint8 [60,60,25] for gray / depth / silhouette,
`frames`, `bbs`), numpy scalars (`label`, `videoId`, `gait`, `compressFactor`, `cam`) become attributes of the root group.
`load_sample` returns them as one dict, as `dd.io.load` does (`data/mj_dataGeneratorMMUWYHsingle_repetitions.py:292-295`);
`stack_raw` lines the `data` arrays of a batch up in the layout `batching.DeviceBatchAssembler` uploads.
"""
from __future__ import annotations

import numpy as np

from . import h5lite

_BOOKKEEPING = ("CLASS", "VERSION", "TITLE", "FLAVOR", "PYTABLES_FORMAT_VERSION", "FILTERS", "DEEPDISH_IO_VERSION",
                "DEEPDISH_IO_UNPACK", "DEEPDISH_IO_ROOT_IS_SNS")


def load_sample(path):
    """dict of a sample file's arrays and scalars.  zlib ('deflate' + shuffle) is deepdish's default compression; a file
    compressed with blosc raises h5lite.H5Error naming the filter."""
    f = h5lite.File(path)
    out = {k: v for k, v in f.attrs.items() if k not in _BOOKKEEPING}
    for k in f.keys():
        node = f[k]
        if isinstance(node, h5lite.Dataset):
            a = node.read()
            # deepdish cannot store a zero-sized array as a node: it stores the array's SHAPE (int64) plus the node attribute
            # `zeroarray_dtype`, and dd.io.load rebuilds np.zeros(shape, dtype) -- an empty recording (generateOFData.py:164-176)
            zdt = node.attrs.get("zeroarray_dtype") if hasattr(node.attrs, "get") else None
            if zdt is not None:
                zdt = zdt.decode("ascii") if isinstance(zdt, (bytes, np.bytes_)) else str(zdt)
                a = np.zeros(tuple(int(v) for v in np.asarray(a).reshape(-1)), dtype=np.dtype(zdt))
            out[k] = a
    return out


def stack_raw(samples, channels):
    """[nbase,60,60,25*channels] array of the samples' `data` (int16 for optical flow, uint8 otherwise); an empty `data`
    (a recording without detections, generateOFData.py:164-176) is refused, as the generator drops such files (:118-146)."""
    want = (60, 60, 25 * channels)
    arrs = []
    for i, s in enumerate(samples):
        a = np.asarray(s["data"])
        if a.shape != want:
            raise ValueError("sample %d: data of shape %r, expected %r" % (i, a.shape, want))
        arrs.append(a)
    return np.stack(arrs)
