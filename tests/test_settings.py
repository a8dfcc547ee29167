"""Settings / arithmetic resolution of a GaitCore without a GPU: which arithmetic a model gets when nobody names one and
the launch switches exclude the path the default arithmetic runs on."""
import pytest

from ugaitnet_amd.config import Settings


def test_default_arithmetic_is_x3():
    s = Settings.from_env({})
    assert s.conv_precision == "f32x3" and s.x3_ok() and not s.precision_named
    assert s.resolve_precision() == ("f32x3", False)
    assert s.resolve_precision("bf16") == ("bf16", False)


@pytest.mark.parametrize("switch", [{"UGN_WINO": "0"}, {"UGN_PAIR": "0"}, {"UGN_MERGE": "0"}, {"UGN_A1_BITS": "0"}, {"UGN_ROUTED": "1"}])
def test_fp32_kernel_switches_without_a_named_arithmetic_fall_back_to_f32(switch):
    s = Settings.from_env(switch)
    assert not s.x3_ok()
    assert s.resolve_precision() == ("f32", True)           # (GaitCore warns once and builds the fp32-MFMA model)
    # a NAMED arithmetic is never replaced: the environment's, or the constructor argument
    named = Settings.from_env(dict(switch, UGN_CONV_PRECISION="f32x3"))
    assert named.precision_named and named.resolve_precision() == ("f32x3", False)      # -> GaitCore raises ValueError for it
    assert s.resolve_precision("f32x3") == ("f32x3", False)
    assert Settings.from_env(dict(switch, UGN_CONV_PRECISION="f32")).resolve_precision() == ("f32", False)


def test_replace_keeps_its_own_copy():
    a = Settings.from_env({})
    b = a.replace(conv_precision="f32", persistent_wgs=64)
    assert (a.conv_precision, a.persistent_wgs) == ("f32x3", 0) and (b.conv_precision, b.persistent_wgs) == ("f32", 64)
