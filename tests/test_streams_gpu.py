"""The backward overlaps weight gradients and data gradients on two streams (and optionally more): every stream mode must
produce bit-identical parameter trajectories, otherwise a launch would be reading data another one is still writing."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_stream_modes_give_identical_training_trajectories(dev):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_streams.py"), "4"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical trajectories" in r.stdout
