"""Writes the small HDF5 fixtures under tests/golden/h5/ with the real HDF5 library (h5py), so that ugaitnet_amd/h5lite.py
is checked against files it did not write itself.

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py        (h5py 3.3 / libhdf5 1.10; NOT the interpreter the tests run on)

* keras_weights_small.h5  -- the layout of keras `save_weights` (tf.keras 2.4 `save_weights_to_hdf5_group`): root attributes
  layer_names / backend / keras_version, one group per layer with a weight_names attribute and the datasets below
  <layer>/<weight name>; layer and weight names as the reference's gaitset graph auto-generates them
  (nets/mj_uwyhNets_ba.py:428-482), tiny shapes.
* keras_model_small.h5    -- the same below /model_weights, as `model.save` writes it (plus model_config / training_config).
* dd_sample_of.h5         -- the layout deepdish/PyTables give one optical-flow sample (data/generateOFData.py:137-149):
  chunked + shuffle + deflate int16 array `data` [60,60,50], small arrays, numpy scalars as root attributes.
* features.h5             -- format features: nested groups, >8 and >64 entries in a group, scalar / empty / big-endian /
  float64 / int8 datasets, chunked datasets with edge chunks and fletcher32, fixed- and variable-length string attributes.
* latest.h5              -- the same kind of content written with libver='latest' (what the reader takes of it, and what it refuses).
The expected contents are stored beside each file as <name>.npz by this script (read back through h5py).
"""
import os
import sys

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(2323)


def keras_layers(nmod=2, small=True):
    """(layer name, [(weight name, shape)]) in model.layers order for a fresh-process gaitset model of nmod modalities."""
    c = lambda *s: tuple(max(1, d // 8) if small and i >= 2 else d for i, d in enumerate(s))
    layers = []
    for m in range(nmod):
        cin = 2 if m == 0 else 1
        td = lambda i: "time_distributed" + ("_%d" % (15 * m + i) if 15 * m + i else "")
        cv = lambda i: "conv2d" + ("_%d" % (10 * m + i) if 10 * m + i else "")
        mm = "mat_mul" + ("_%d" % m if m else "")
        layers += [
            (td(1), [(td(1) + "/kernel:0", (5, 5, cin, c(0, 0, 32)[2]))]),
            (td(3), [(td(3) + "/kernel:0", c(3, 3, 32, 32))]),
            (cv(2), [(cv(2) + "/kernel:0", c(3, 3, 32, 64))]),
            (cv(3), [(cv(3) + "/kernel:0", c(3, 3, 64, 64))]),
            (td(6), [(td(6) + "/kernel:0", c(3, 3, 32, 64))]),
            (td(8), [(td(8) + "/kernel:0", c(3, 3, 64, 64))]),
            (cv(6), [(cv(6) + "/kernel:0", c(3, 3, 64, 128))]),
            (cv(7), [(cv(7) + "/kernel:0", c(3, 3, 128, 128))]),
            (td(11), [(td(11) + "/kernel:0", c(3, 3, 64, 128))]),
            (td(13), [(td(13) + "/kernel:0", c(3, 3, 128, 128))]),
            (mm, [("MatMul_kernel[%d]:0" % (17 + m), (62, 4, 8) if small else (62, 128, 256))]),
            (td(2), []), ("lambda" + ("_%d" % m if m else ""), []),
        ]
    layers += [("fusion", []), ("signature", []), ("flatten", []),
               ("classprob", [("classprob/kernel:0", (62 * (8 if small else 256), 6)), ("classprob/bias:0", (6,))])]
    return layers


def write_keras_weights(g, layers, expect, prefix=""):
    g.attrs["layer_names"] = [n.encode("utf8") for n, _ in layers]
    g.attrs["backend"] = "tensorflow".encode("utf8")
    g.attrs["keras_version"] = "2.4.0".encode("utf8")
    for name, ws in layers:
        lg = g.create_group(name)
        lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in ws]
        for w, shape in ws:
            val = rng.standard_normal(shape).astype(np.float32)
            d = lg.create_dataset(w, val.shape, dtype=val.dtype)
            d[:] = val
            expect[prefix + name + "/" + w] = val


def main():
    layers = keras_layers()
    exp = {}
    with h5py.File(os.path.join(OUT, "keras_weights_small.h5"), "w") as f:
        write_keras_weights(f, layers, exp)
    np.savez_compressed(os.path.join(OUT, "keras_weights_small.npz"), **exp)

    exp = {}
    with h5py.File(os.path.join(OUT, "keras_model_small.h5"), "w") as f:
        f.attrs["keras_version"] = "2.4.0".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = '{"class_name": "Functional", "config": {"name": "model"}}'.encode("utf8")
        f.attrs["training_config"] = '{"loss": null}'.encode("utf8")
        write_keras_weights(f.create_group("model_weights"), layers, exp, "model_weights/")
    np.savez_compressed(os.path.join(OUT, "keras_model_small.npz"), **exp)

    # deepdish / PyTables sample: smooth int16 flow so that the fixture stays small after deflate
    yy, xx, tt = np.meshgrid(np.arange(60), np.arange(60), np.arange(50), indexing="ij")
    data = np.int16(np.round(900 * np.sin(0.11 * yy + 0.07 * tt) * np.cos(0.09 * xx)) + (tt % 2) * 25)
    frames = np.uint16(np.arange(31, 56))
    bbs = np.uint8(rng.integers(0, 255, (25, 4)))
    with h5py.File(os.path.join(OUT, "dd_sample_of.h5"), "w") as f:
        for k, v in (("CLASS", "GROUP"), ("PYTABLES_FORMAT_VERSION", "2.1"), ("TITLE", ""), ("VERSION", "1.0")):
            f.attrs[k] = np.bytes_(v)
        f.attrs["DEEPDISH_IO_VERSION"] = np.int64(12)
        f.attrs["label"] = np.uint16(42)
        f.attrs["videoId"] = np.uint16(1234)
        f.attrs["gait"] = np.uint8(2)
        f.attrs["compressFactor"] = np.uint8(100)
        f.attrs["cam"] = np.int64(90)
        d = f.create_dataset("data", data=data, chunks=(15, 30, 25), compression="gzip", compression_opts=5, shuffle=True)
        for k, v in (("CLASS", "CARRAY"), ("TITLE", ""), ("VERSION", "1.1")):
            d.attrs[k] = np.bytes_(v)
        for name, arr in (("frames", frames), ("bbs", bbs)):
            d = f.create_dataset(name, data=arr)
            for k, v in (("CLASS", "ARRAY"), ("FLAVOR", "numpy"), ("TITLE", ""), ("VERSION", "2.4")):
                d.attrs[k] = np.bytes_(v)
    np.savez_compressed(os.path.join(OUT, "dd_sample_of.npz"), data=data, frames=frames, bbs=bbs, label=np.uint16(42),
             videoId=np.uint16(1234), gait=np.uint8(2), compressFactor=np.uint8(100), cam=np.int64(90))

    exp = {}
    with h5py.File(os.path.join(OUT, "features.h5"), "w") as f:
        g = f.create_group("a/b/c")
        exp["a/b/c/x"] = np.arange(24, dtype=np.float64).reshape(2, 3, 4)
        g.create_dataset("x", data=exp["a/b/c/x"])
        for i in range(100):     # a group that needs several symbol-table nodes
            exp["many/d%03d" % i] = np.array([i, -i], np.int32)
            f.create_dataset("many/d%03d" % i, data=exp["many/d%03d" % i])
        exp["scalar"] = np.float32(3.25)
        f.create_dataset("scalar", data=exp["scalar"])
        exp["empty"] = np.zeros((0,), np.int16)
        f.create_dataset("empty", data=exp["empty"])
        exp["big_endian"] = np.arange(7, dtype=np.int32)
        f.create_dataset("big_endian", data=np.arange(7, dtype=">i4"))
        exp["i8"] = np.int8([-128, 0, 127])
        f.create_dataset("i8", data=exp["i8"])
        exp["chunked_edge"] = rng.integers(0, 1000, (37, 23)).astype(np.uint16)
        f.create_dataset("chunked_edge", data=exp["chunked_edge"], chunks=(16, 10), compression="gzip", shuffle=True,
                         fletcher32=True)
        exp["chunked_plain"] = rng.standard_normal((9, 5, 4)).astype(np.float32)
        f.create_dataset("chunked_plain", data=exp["chunked_plain"], chunks=(4, 5, 3))
        exp["never_written"] = np.zeros((3, 2), np.float32)
        f.create_dataset("never_written", (3, 2), dtype=np.float32)
        exp["compact_like"] = np.uint8([1, 2, 3])
        f.create_dataset("compact_like", data=exp["compact_like"])
        f.attrs["fixed"] = np.bytes_("fixed-length")
        f.attrs["vlen_bytes"] = b"variable bytes"
        f.attrs["vlen_str"] = "variable str é"
        f.attrs["list_fixed"] = [b"ab", b"cdef", b""]
        f.attrs["floats"] = np.float64([1.5, -2.5])
        f.attrs["int_scalar"] = np.int64(-7)
        f["a"].attrs["on_group"] = np.uint8(9)
        f["scalar"].attrs["on_dataset"] = np.float32(0.5)
        f.attrs["vlen_list"] = np.array(["x", "yy", "zzz"], dtype=h5py.string_dtype())
    np.savez_compressed(os.path.join(OUT, "features.npz"), **exp)
    # libver='latest': version-2 superblock and object headers, compact link messages; dense groups and version-4 chunk
    # indexes are outside the reader's subset and must be refused by name
    with h5py.File(os.path.join(OUT, "latest.h5"), "w", libver="latest") as f:
        f.attrs["layer_names"] = [b"a", b"bb"]
        g = f.create_group("a")
        g.attrs["weight_names"] = [b"a/kernel:0"]
        g.create_dataset("a/kernel:0", data=np.arange(12, dtype=np.float32).reshape(3, 4))
        f.create_dataset("chunked", data=np.arange(100, dtype=np.int16).reshape(10, 10), chunks=(4, 4), compression="gzip")
        for i in range(20):
            f.create_dataset("many/d%d" % i, data=np.float32(i))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    sys.exit(main())
