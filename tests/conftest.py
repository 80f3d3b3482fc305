import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The suites need libautoinst_hip.so; build it (hipcc cross-compiles without a GPU) if it is missing."""
    lib = os.path.join(ROOT, "autoinst_amd", "libautoinst_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    yield


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("g") and f.endswith(".npz"))


@pytest.fixture(scope="session")
def ctx():
    from autoinst_amd import ncuts_api
    return ncuts_api.default_context()
