import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "stubccl")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # run-time kernel instantiation (tensorbnn_amd/jit.py) only where a test asks for it (jit=True)
    os.environ.setdefault("TBNN_JIT", "0")
    # the C-ABI library is built in-tree (hipcc cross-compiles gfx950 without a GPU); build it if absent/stale
    try:
        from tensorbnn_amd import build as _b
        _b.build(force=False, verbose=False)
    except Exception as e:          # the tests that need it will fail loudly
        print("libtbnn build failed:", e)


def _has_gpu():
    try:
        from tensorbnn_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: the product has no fallback.
    pass


@pytest.fixture(scope="session")
def native():
    from tensorbnn_amd import _native
    return _native
