import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "solstrale-rust_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Everything is built in-tree before any test: HIP library + host library (hipcc cross-compiles without a GPU) and
    the oracle. On the GPU box the prebuilt .so files travel with the snapshot and are only rebuilt when stale."""
    import __graft_entry__
    __graft_entry__.build()
    yield
