"""model-config.hdf5 (ugaitnet_amd/ddconfig.py): the deepdish layout of the reference's architecture dictionary
(mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:474-489), written and read back without deepdish / PyTables.
Parity unpinned (no deepdish-written file exists here): these tests pin the layout the module documents and the round trip."""
import numpy as np
import pytest

from ugaitnet_amd import ddconfig, h5lite


def _modelpars():
    # the dictionary of the CASIA-B main, two-modality gaitset run (input_shape is a list of tuples there)
    return {"filters_size": [7, 5, 3, 2], "filters_numbers": [96, 192, 512, 4096],
            "input_shape": [(25, 60, 60, 2), (25, 60, 60, 1)], "ndense_units": 0, "weight_decay": 1e-4, "dropout": 0.4,
            "optimizer": "Adam", "margin": 0.2, "custom": "TripletSemiHardLoss", "nclasses": 74, "softlabel": 0,
            "use3D": False, "loss_weights": [1.0, 0.1], "fMerge": "sign_max"}


def test_round_trip_keeps_values_and_types(tmp_path):
    cfg = _modelpars()
    cfg.update(none=None, nested={"a": 1, "b": [1.5, "x", (2, 3)]}, arr=np.arange(6, dtype=np.float32).reshape(2, 3),
               empty=np.zeros((0, 4), np.int16), flag=True, npint=np.int32(7), npfloat=np.float32(0.5))
    p = str(tmp_path / "model-config.hdf5")
    ddconfig.save(p, cfg)
    got = ddconfig.load(p)
    assert set(got) == set(cfg)
    for k, want in cfg.items():
        if isinstance(want, np.ndarray):
            assert got[k].dtype == want.dtype and got[k].shape == want.shape and np.array_equal(got[k], want), k
        elif isinstance(want, (np.integer, np.floating)):
            assert got[k] == want, k
        else:
            assert got[k] == want and type(got[k]) is type(want), (k, got[k], want)
    assert isinstance(got["input_shape"][0], tuple) and isinstance(got["filters_size"], list)


def test_layout_is_the_documented_one(tmp_path):
    p = str(tmp_path / "c.hdf5")
    ddconfig.save(p, _modelpars())
    f = h5lite.File(p)
    # scalars: attributes of the root group, typed as PyTables types Python scalars
    assert int(f.attrs["DEEPDISH_IO_VERSION"]) == 12
    assert f.attrs["nclasses"].dtype == np.int64 and f.attrs["margin"].dtype == np.float64
    assert f.attrs["use3D"].dtype == np.bool_ and not bool(f.attrs["use3D"])
    assert bytes(f.attrs["optimizer"]) == b"Adam" and bytes(f.attrs["fMerge"]) == b"sign_max"
    # sequences: groups titled kind:N, elements i0 .. i{N-1}, scalar elements as attributes of that group
    assert sorted(f.keys()) == ["filters_numbers", "filters_size", "input_shape", "loss_weights"]
    g = f["filters_size"]
    assert bytes(g.attrs["TITLE"]) == b"list:4" and [int(g.attrs["i%d" % i]) for i in range(4)] == [7, 5, 3, 2]
    s = f["input_shape"]
    assert bytes(s.attrs["TITLE"]) == b"list:2" and sorted(s.keys()) == ["i0", "i1"]
    assert bytes(s["i1"].attrs["TITLE"]) == b"tuple:4" and int(s["i1"].attrs["i3"]) == 1


def test_non_dictionary_top_level_unpacks(tmp_path):
    p = str(tmp_path / "v.hdf5")
    ddconfig.save(p, [1, 2.5, ("a", None)])
    assert ddconfig.load(p) == [1, 2.5, ("a", None)]
    assert bool(h5lite.File(p).attrs["DEEPDISH_IO_UNPACK"])


def test_what_cannot_be_stored_natively_is_refused(tmp_path):
    p = str(tmp_path / "bad.hdf5")
    with pytest.raises(ValueError, match="fMerge"):
        ddconfig.save(p, {"fMerge": max})                 # a callable: deepdish would pickle it
    with pytest.raises(ValueError, match="keys"):
        ddconfig.save(p, {"d": {1: 2}})
    # a pickled attribute in a file (what PyTables writes for objects it cannot type) is named, not unpickled
    w = h5lite.Writer()
    w.set_attr("", "fMerge", b"\x80\x04\x95\x0c\x00\x00\x00")
    w.save(p)
    with pytest.raises(ValueError, match="fMerge"):
        ddconfig.load(p)
    # unknown node kinds likewise
    w = h5lite.Writer()
    w.create_group("obj")
    w.set_attr("obj", "TITLE", "pyobject:")
    w.save(p)
    with pytest.raises(ValueError, match="obj"):
        ddconfig.load(p)
