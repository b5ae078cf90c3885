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


@pytest.fixture(autouse=True, scope="session")
def _visited_format_from_env():
    """DANN_TEST_VISITED_FORMAT=16 (with DANN_TUNE_OFF=4: no teams) runs every GPU test with 16-bit visited-table
    entries wherever the kernels have them -- results never depend on the table, so the whole suite must stay green."""
    fmt = int(os.environ.get("DANN_TEST_VISITED_FORMAT", "0") or 0)
    if not fmt:
        yield
        return
    import diskann_amd as da
    orig = da.Provider.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        self.set_visited_format(fmt)

    da.Provider.__init__ = init
    yield
    da.Provider.__init__ = orig
