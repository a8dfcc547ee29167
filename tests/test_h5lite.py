"""ugaitnet_amd/h5lite.py against files written by the real HDF5 library (tests/golden/h5, made with h5py by
tests/golden/make_h5_fixtures.py), the Keras-checkpoint mapping on top of it, and the writer (read back by h5lite here and,
where an interpreter with h5py exists on the machine, by the HDF5 library itself)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from ugaitnet_amd import h5lite, keras_h5, samples

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")


def _same(got, exp):
    got = np.asarray(got)
    return got.dtype == exp.dtype and got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("name", ["features", "keras_weights_small", "keras_model_small"])
def test_reader_matches_h5py_on_every_dataset(name):
    f = h5lite.File(os.path.join(H5, name + ".h5"))
    exp = np.load(os.path.join(H5, name + ".npz"))
    assert len(exp.files) > 0
    for k in exp.files:
        assert _same(f[k].read(), exp[k]), k
    listed = {p for p, _ in f.visit_datasets()}
    assert listed == set(exp.files)


def test_reader_features():
    f = h5lite.File(os.path.join(H5, "features.h5"))
    assert len(f["many"]) == 100 and f["many"].keys()[0] == "d000"       # group spread over several symbol-table nodes
    assert f["a/b/c/x"].shape == (2, 3, 4) and f["a"]["b"]["c"]["x"].dtype == np.float64
    assert f["scalar"].shape == () and f["empty"].shape == (0,)
    a = f.attrs
    assert a["fixed"] == b"fixed-length" and a["vlen_bytes"] == b"variable bytes" and a["vlen_str"] == "variable str é"
    assert list(a["vlen_list"]) == ["x", "yy", "zzz"] and list(a["list_fixed"]) == [b"ab", b"cdef", b""]
    assert a["int_scalar"] == -7 and a["floats"].tolist() == [1.5, -2.5]
    assert f["a"].attrs["on_group"] == 9 and f["scalar"].attrs["on_dataset"] == 0.5
    assert "nope" not in f
    with pytest.raises(KeyError):
        f["a/b/nope"]


def test_reader_refuses_what_it_does_not_understand(tmp_path):
    p = tmp_path / "not.h5"
    p.write_bytes(b"PK\x03\x04 this is a zip")
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p))
    raw = open(os.path.join(H5, "dd_sample_of.h5"), "rb").read()
    (tmp_path / "cut.h5").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(Exception):     # (struct.error or H5Error, depending on where the file ends)
        samples.load_sample(str(tmp_path / "cut.h5"))["data"]


def test_libver_latest_files():
    f = h5lite.File(os.path.join(H5, "latest.h5"))           # version-2 superblock / object headers, compact links
    assert f.keys() == ["a", "chunked", "many"] and list(f.attrs["layer_names"]) == [b"a", b"bb"]
    assert np.array_equal(f["a/a/kernel:0"].read(), np.arange(12, dtype=np.float32).reshape(3, 4))
    assert list(f["a"].attrs["weight_names"]) == [b"a/kernel:0"]
    with pytest.raises(h5lite.H5Error, match="fractal heap"):
        f["many"].keys()
    with pytest.raises(h5lite.H5Error, match="version-4 chunked"):
        f["chunked"].read()


def test_deepdish_sample():
    s = samples.load_sample(os.path.join(H5, "dd_sample_of.h5"))
    exp = np.load(os.path.join(H5, "dd_sample_of.npz"))
    assert set(s) == set(exp.files)                       # PyTables / deepdish bookkeeping attributes are dropped
    for k in exp.files:
        assert _same(s[k], exp[k]), k
    raw = samples.stack_raw([s, s], channels=2)
    assert raw.shape == (2, 60, 60, 50) and raw.dtype == np.int16
    with pytest.raises(ValueError):
        samples.stack_raw([s], channels=1)


def test_keras_weights_map_onto_parameters():
    exp = np.load(os.path.join(H5, "keras_weights_small.npz"))
    for fname, pre in (("keras_weights_small.h5", ""), ("keras_model_small.h5", "model_weights/")):
        e = np.load(os.path.join(H5, fname[:-3] + ".npz"))
        layers = keras_h5.read_layers(os.path.join(H5, fname))
        assert [n for n, _ in layers][:3] == ["time_distributed_1", "time_distributed_3", "conv2d_2"]
        got = keras_h5.assign(layers, nmod=2, nclasses=6)
        assert len(got) == 2 * 11 + 2
        for mi in range(2):
            for pname in ("a1", "a2", "a3", "a4", "a5", "a6", "b1", "b2", "b3", "b4"):
                ln = keras_h5.keras_layer_name(mi, pname)
                assert np.array_equal(got["m%d.%s" % (mi, pname)], e[pre + ln + "/" + ln + "/kernel:0"]), (mi, pname)
            mm = keras_h5.keras_layer_name(mi, "fc")
            assert np.array_equal(got["m%d.fc" % mi], e["%s%s/MatMul_kernel[%d]:0" % (pre, mm, 17 + mi)])
        assert got["head.wc"].shape == (62 * 8, 6) and got["head.bc"].shape == (6,)
    assert got["m0.a1"].shape[:3] == (5, 5, 2) and got["m1.a1"].shape[:3] == (5, 5, 1)
    assert exp is not None


def _params(rng, in_channels, ncls):
    from ugaitnet_amd.engine import branch_param_shapes
    p = {}
    for mi, c in enumerate(in_channels):
        for n, s in branch_param_shapes(c):
            p["m%d.%s" % (mi, n)] = rng.standard_normal(s).astype(np.float32)
    p["head.wc"] = rng.standard_normal((62 * 256, ncls)).astype(np.float32)
    p["head.bc"] = rng.standard_normal((ncls,)).astype(np.float32)
    return p


def test_writer_round_trip_full_size(tmp_path):
    """A real-size 2-modality checkpoint through write_weights -> read_layers -> assign; both layouts; shifted counters."""
    rng = np.random.default_rng(4)
    p = _params(rng, (2, 1), 5)
    for below in ("", "model_weights"):
        path = str(tmp_path / ("w_%s.hdf5" % (below or "root")))
        keras_h5.write_weights(path, p, (2, 1), 5, below=below,
                               extra={"optimizer_weights/iterations": np.array([7], np.int64), "ugaitnet_config": json.dumps({"a": 1})})
        got = keras_h5.assign(keras_h5.read_layers(path), 2, 5)
        assert set(got) == set(p)
        assert all(np.array_equal(got[k], p[k]) for k in p)
        f = h5lite.File(path)
        assert f["optimizer_weights/iterations"].read().tolist() == [7] and json.loads(f.attrs["ugaitnet_config"]) == {"a": 1}
    # a model built as the SECOND model of a process: every automatic name is shifted, the order is not
    w = h5lite.Writer()
    names = []
    for mi in range(2):
        for pname in ("a1", "a2", "b1", "b2", "a3", "a4", "b3", "b4", "a5", "a6", "fc"):
            ln = keras_h5.keras_layer_name(mi + 2, pname)
            names.append(ln)
            w.create_dataset("%s/%s/kernel:0" % (ln, ln), p["m%d.%s" % (mi, pname)])
            w.set_attr(ln, "weight_names", [("%s/kernel:0" % ln).encode()])
    w.set_attr("", "layer_names", [n.encode() for n in names])
    w.save(str(tmp_path / "shifted.h5"))
    got = keras_h5.assign(keras_h5.read_layers(str(tmp_path / "shifted.h5")), 2, 5)
    assert all(np.array_equal(got[k], p[k]) for k in got) and len(got) == 22


def test_writer_output_is_read_by_the_hdf5_library(tmp_path):
    """Only where an interpreter with h5py exists (it does on the build image: /opt/conda/bin/python3.9)."""
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py on this machine")
    rng = np.random.default_rng(5)
    w = h5lite.Writer()
    exp = {}
    for i in range(70):     # more than one symbol-table node, names that sort differently from their creation order
        k = "g%d/layer_%d/kernel:0" % (i % 3, 70 - i)
        exp[k] = rng.standard_normal((2, 3)).astype(np.float32)
        w.create_dataset(k, exp[k])
    exp["i16"] = np.int16([[1, -2], [3, 4]])
    exp["f64"] = np.float64([0.25])
    exp["u8s"] = np.uint8(200)
    for k in ("i16", "f64", "u8s"):
        w.create_dataset(k, exp[k])
    w.set_attr("", "layer_names", [b"alpha", b"be", b"gamma_long_name"])
    w.set_attr("g1", "count", np.int32(3))
    w.set_attr("i16", "unit", "px")
    path = str(tmp_path / "lib.h5")
    w.save(path)
    np.savez(str(tmp_path / "exp.npz"), **exp)
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r'); e = np.load(sys.argv[2])\n"
            "assert all(np.array_equal(f[k][()], e[k]) and f[k].dtype == e[k].dtype and f[k].shape == e[k].shape for k in e.files)\n"
            "assert list(f.attrs['layer_names']) == [b'alpha', b'be', b'gamma_long_name']\n"
            "assert f['g1'].attrs['count'] == 3 and f['i16'].attrs['unit'] == b'px'\n"
            "n = []; f.visit(n.append); assert sum(isinstance(f[k], h5py.Dataset) for k in n) == len(e.files)\n"
            "print('library read ok')\n")
    r = subprocess.run([py, "-c", code, path, str(tmp_path / "exp.npz")], capture_output=True, text=True)
    assert r.returncode == 0 and "library read ok" in r.stdout, r.stderr[-1500:]
