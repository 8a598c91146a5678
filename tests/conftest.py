import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def fixture_cloud():
    """The reference's shipped data file as committed arrays (tests/golden/make_fixtures.py)."""
    import numpy as np

    xyz = np.load(os.path.join(GOLDEN, "intersection00056_xyz.npy"), allow_pickle=False)
    xyzn = np.load(os.path.join(GOLDEN, "intersection00056_xyzn.npy"), allow_pickle=False)
    return xyz, xyzn


@pytest.fixture(scope="session")
def lom():
    """The product package bound to the HIP C-ABI library (fails loudly if not built)."""
    import lidar_odometry_demo_amd as pkg

    pkg.capi.lib()
    return pkg


@pytest.fixture
def counted_search(monkeypatch):
    """Handles created inside the test produce the reference-ALGORITHM counts (LOM_COUNT_CANDIDATES=1 at create =
    LOM_OPT_COUNT_CANDIDATES: all 27 slots per query) -- for the tests that compare counters [28..30] of the reduced sums
    or cand_total / occ_total with the oracle's.  tests/test_gpu_parity.py runs every test both ways; rank processes
    spawned by a test inherit the variable."""
    monkeypatch.setenv("LOM_COUNT_CANDIDATES", "1")
