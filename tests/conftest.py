import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "pde-solver_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libpdegpu.so, built in-tree by __graft_entry__.build() / make."""
    import __graft_entry__ as ge
    from pdegpu import _lib
    if not _lib.LIB_PATH.exists():
        ge.build()
    return _lib.load_library()


@pytest.fixture(scope="session")
def gpu_ctx(lib):
    from pdegpu import _lib
    return _lib.context(0)
