import os
import sys

import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before torch initialises HIP (geot_amd/__init__.py explains)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library with hipcc (it
    cross-compiles without a GPU) instead of failing every test on a missing file.  A build error is left for
    the tests to report -- the product itself never falls back to anything."""
    from geot_amd import build as hip_build
    try:
        hip_build.build()      # mtime-incremental: also refreshes a stale library left by an older checkout
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("geot_amd: building %s failed: %s\n" % (hip_build.LIB, e))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (C restatement, built on demand).  Checker only."""
    from oracle import capi
    capi.build()
    return capi


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
