import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "h2: needs a library built with the opt-in f16x2 kernel set (python -m ugaitnet_amd.build --h2)")


FULLSIZE = "test_fullsize_parity_gpu"


def pytest_collection_modifyitems(config, items):
    """Without a GPU every test that carries the `gpu` marker is skipped, whether or not it takes the `dev` fixture.
    With one: the full-size parity tests are placed inside the session so that their fp64 evaluations overlap the other GPU tests."""
    import re
    import torch
    if torch.cuda.is_available():
        # the f16x2 ("h2") kernel set is an opt-in build since round 6 (python -m ugaitnet_amd.build --h2): its tests -- the two h2
        # modules and every case whose id names h2 -- run only against a library that carries it
        from ugaitnet_amd import _lib
        if not _lib.has_h2():
            no_h2 = pytest.mark.skip(reason="libugaitnet_hip.so was built without the opt-in f16x2 set (build --h2)")
            for item in items:
                if "h2" in item.keywords or re.search(r"test_mm_gpu\.py|test_h2_elem_gpu\.py|h2|f16x2", item.nodeid):
                    item.add_marker(no_h2)
        # The full-size parity cases run INSIDE the session, not first: their fp64 oracle evaluations (host cores only) run on a child
        # process from the session's start (`_oracle_ahead` below), and the forced-routing evaluations of the default-arithmetic cases
        # (20-110 s of fp64 each, one at a time on a background thread) run beside the tests that follow them.  Order: 40 % of the
        # other tests | C3, C2 | the rest of the other tests | C4 and the other arithmetics | the test that joins the forced evaluations.
        full = [i for i in items if FULLSIZE in i.nodeid]
        rest = [i for i in items if FULLSIZE not in i.nodeid]
        early = [i for i in full if hasattr(i, "callspec") and i.callspec.params.get("name") in ("C3", "C2")]
        join = [i for i in full if "forced_to_the_hip_routing" in i.nodeid]
        late = [i for i in full if i not in early and i not in join]
        cut = (2 * len(rest)) // 5 if early else len(rest)
        items[:] = rest[:cut] + early + rest[cut:] + late + join
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _host_threads():
    """torch's CPU thread pool sized to the cores this process may really use (tests/cpu_share.py), minus what the full-size oracle
    child takes while it runs: the numpy / torch-CPU oracles of the tests otherwise start one thread per LOGICAL CPU of the machine."""
    import torch
    from tests.cpu_share import usable_cores
    n = usable_cores()
    torch.set_num_threads(max(2, n if n <= 8 else n // 2))
    yield


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
