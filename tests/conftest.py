import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


FULLSIZE = "test_fullsize_parity_gpu"


def pytest_collection_modifyitems(config, items):
    """Without a GPU every test that carries the `gpu` marker is skipped, whether or not it takes the `dev` fixture.
    With one: the full-size parity tests go to the END of the session -- their fp64 oracle evaluations (host cores only) run on a
    child process from the session's start, beside the other GPU tests (`_oracle_ahead` below)."""
    import re
    import torch
    if torch.cuda.is_available():
        # the f16x2 ("h2") kernel set is an opt-in build since round 6 (python -m ugaitnet_amd.build --h2): its tests -- the two h2
        # modules and every case whose id names h2 -- run only against a library that carries it
        from ugaitnet_amd import _lib
        if not _lib.has_h2():
            no_h2 = pytest.mark.skip(reason="libugaitnet_hip.so was built without the opt-in f16x2 set (build --h2)")
            for item in items:
                if re.search(r"test_mm_gpu\.py|test_h2_elem_gpu\.py|h2|f16x2", item.nodeid):
                    item.add_marker(no_h2)
        items[:] = [i for i in items if FULLSIZE not in i.nodeid] + [i for i in items if FULLSIZE in i.nodeid]
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from ugaitnet_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing: there is no fallback
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _oracle_ahead(request):
    """Starts the fp64-oracle process of tests/test_fullsize_parity_gpu.py for the workloads this session selected."""
    import torch
    cases = []
    for item in request.session.items:
        if FULLSIZE in item.nodeid and hasattr(item, "callspec") and not any(m.name == "skip" for m in item.iter_markers()):
            cases.append(item.callspec.params["name"])
    if cases and torch.cuda.is_available():
        from tests import test_fullsize_parity_gpu as M
        wanted = []
        for n in cases:
            if M.WORKLOAD[n] not in wanted:
                wanted.append(M.WORKLOAD[n])
            M.LAST_SELECTED[M.WORKLOAD[n]] = n
        M.DEFER_FORCED = any("forced_to_the_hip_routing" in item.nodeid for item in request.session.items)
        M.PREFETCH.start(wanted)
    yield
    if cases and torch.cuda.is_available():
        M.PREFETCH.close()
