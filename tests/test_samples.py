"""Sample-file conventions that need no GPU: deepdish's zero-sized arrays, shape checks of the batch loader."""
import os

import numpy as np
import pytest

from ugaitnet_amd import h5lite, samples


def test_deepdish_zero_sized_array_is_rebuilt(tmp_path):
    """deepdish stores np.int16([]) as its shape (int64) + the node attribute zeroarray_dtype; dd.io.load returns zeros(shape).
    The generator must see an EMPTY `data` there (such records are dropped, data/...repetitions.py:118-146)."""
    w = h5lite.Writer()
    w.create_dataset("data", np.array([0], np.int64))
    w.set_attr("data", "zeroarray_dtype", np.bytes_(b"<i2"))
    w.set_attr("", "label", np.uint16(7))
    p = str(tmp_path / "empty_dd.h5")
    w.save(p)
    s = samples.load_sample(p)
    assert s["data"].shape == (0,) and s["data"].dtype == np.int16 and len(s["data"]) == 0
    assert int(s["label"]) == 7
    w = h5lite.Writer()
    w.create_dataset("data", np.array([60, 60, 0], np.int64))
    w.set_attr("data", "zeroarray_dtype", "|u1")
    p2 = str(tmp_path / "empty_dd2.h5")
    w.save(p2)
    assert samples.load_sample(p2)["data"].shape == (60, 60, 0)


def test_stack_raw_refuses_wrong_shapes():
    good = dict(data=np.zeros((60, 60, 50), np.int16))
    assert samples.stack_raw([good, good], 2).shape == (2, 60, 60, 50)
    with pytest.raises(ValueError):
        samples.stack_raw([good, dict(data=np.zeros((1,), np.int64))], 2)
