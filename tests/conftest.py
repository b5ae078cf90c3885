import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A clean checkout has no built artefacts (they are git-ignored): compile libdann_hip.so once so that the ABI
    tests can load it.  hipcc cross-compiles gfx950 without a GPU.  The product itself never builds on import."""
    lib = os.path.join(ROOT, "diskann_amd", "libdann_hip.so")
    if not os.path.exists(lib):
        from diskann_amd import build as hip_build
        hip_build.build()
