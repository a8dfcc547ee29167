"""The C-ABI library loads on a machine without a GPU and exports every symbol include/ugaitnet_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ugaitnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ugn_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_loads():
    from ugaitnet_amd import _lib
    lib = _lib.load()
    assert lib.ugn_abi_version() == 1


def test_every_declared_symbol_is_exported_and_bound():
    from ugaitnet_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libugaitnet_hip.so does not export %s" % n
    assert sorted(_lib.PROTOTYPES) == names, "ugaitnet_amd/_lib.py prototypes are out of sync with the header"


def test_host_only_entry_points():
    """ugn_triplet_indices_host is pure host code: exercised here without a GPU, bit-exact against the oracle."""
    from oracle import ugaitnet_oracle as O
    from ugaitnet_amd import _lib
    lib = _lib.load()
    for labels in (np.repeat(np.arange(12), 2), np.array([0] * 10 + [1] * 10 + [2] * 4), np.repeat(np.arange(4), 10)):
        lab = labels.astype(np.int32)
        m = lab.size
        hp = np.empty(m * m, np.int32); hn = np.empty(m * m, np.int32)
        kp, kn = ctypes.c_int(), ctypes.c_int()
        rc = lib.ugn_triplet_indices_host(lab.ctypes.data, m, hp.ctypes.data, hn.ctypes.data, ctypes.byref(kp), ctypes.byref(kn))
        assert rc == 0
        rhp, rhn, rkp, rkn = O.triplet_index_lists(labels)
        assert (kp.value, kn.value) == (rkp, rkn)
        assert np.array_equal(hp[:m * rkp], rhp) and np.array_equal(hn[:m * rkn], rhn)
    bad = np.array([0, 0, 0, 1, 1], np.int32)
    hp = np.empty(25, np.int32); hn = np.empty(25, np.int32)
    rc = lib.ugn_triplet_indices_host(bad.ctypes.data, 5, hp.ctypes.data, hn.ctypes.data, ctypes.byref(kp), ctypes.byref(kn))
    assert rc == -22 and b"divisible" in lib.ugn_last_error()


def test_argument_validation_without_gpu():
    from ugaitnet_amd import _lib
    lib = _lib.load()
    assert lib.ugn_conv3x3_fwd(None, None, None, None, 1, 64, 32, 32, 1, None) == -22
    assert lib.ugn_conv3x3_wgrad_ws(600, 64, 32, 32) > 0
    assert lib.ugn_conv3x3_wgrad_ws(600, 48, 32, 32) == 0          # unsupported shape
    assert lib.ugn_conv5x5_in_wgrad_ws(600, 3) == 0


def test_persistent_grid_setting_round_trips_without_gpu():
    """ugn_set_persistent_wgs / ugn_get_persistent_wgs: the library's one process-wide launch setting is plain host state."""
    from ugaitnet_amd import _lib
    lib = _lib.load()
    try:
        assert lib.ugn_get_persistent_wgs() == 256
        assert lib.ugn_set_persistent_wgs(224) == 0 and lib.ugn_get_persistent_wgs() == 224
        assert lib.ugn_set_persistent_wgs(4) == -22 and b"8..256" in lib.ugn_last_error()      # (out of range: refused, unchanged)
        assert lib.ugn_get_persistent_wgs() == 224
    finally:
        assert lib.ugn_set_persistent_wgs(0) == 0 and lib.ugn_get_persistent_wgs() == 256


def test_opt_in_f16x2_entry_points_are_declared_apart():
    """The f16x2 ("H2") set is an opt-in build: its entry points live in include/ugaitnet_hip_h2.h, none of them in the main header,
    and the default library exports them only when it was built with --h2 (then all of them, bound by ugaitnet_amd._lib)."""
    from ugaitnet_amd import _lib
    text = open(os.path.join(ROOT, "include", "ugaitnet_hip_h2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    h2 = sorted(set(re.findall(r"\b(ugn_[a-z0-9_]+)\s*\(", text)))
    assert len(h2) >= 20 and sorted(_lib.PROTOTYPES_H2) == h2
    assert not set(h2) & set(declared_symbols())
    lib = ctypes.CDLL(_lib.LIB_PATH)
    have = [hasattr(lib, n) for n in h2]
    assert all(have) or not any(have)
    assert _lib.has_h2() == all(have)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ugaitnet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle's layout", ""), "%s mentions the oracle" % f


def test_host_side_api_objects():
    from ugaitnet_amd.keras_compat import Average, Maximum, fusion_mode, optimizers, sign_max
    assert fusion_mode(sign_max) == "sign_max" and fusion_mode(Maximum) == "max" and fusion_mode(Average(name="f")) == "avg"
    with pytest.raises(ValueError):
        fusion_mode(lambda name=None: object())
    opt = optimizers.Adam(lr=1e-4)
    assert opt.lr == 1e-4 and opt.epsilon == 1e-7 and opt.beta_2 == 0.999
    from ugaitnet_amd.nets.mj_uwyhNets_ba import UWYHSemiNet
    assert UWYHSemiNet.get_weights_filename("/a/b/model-state-0002.hdf5") == "/a/b/model-state-0002_weights.hdf5"
    assert UWYHSemiNet.get_netconfig_filename("/a/b/model-state-0002.hdf5") == "/a/b/model-config.hdf5"
    with pytest.raises(NotImplementedError):
        UWYHSemiNet.build((25, 60, 60, 1), 4, [7, 5, 3, 2], [96, 192, 512, 4096], gaitset=False)
