import os
import sys

# (before NumPy is imported: the GPU boxes schedule 16 of 256 logical CPUs; a BLAS pool of 256 spinning threads gets the container
# throttled, launch thread included -- tensorbnn_amd._native warns about it, NOTES.md round 4.)  A default, not a requirement: the
# synthetic problems are generated in fp64 and rounded once (oracle.synth_problem), so no input depends on the thread count.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")

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


def wait_gpu_quiet(timeout=30.0):
    """Multi-process tests share ONE card with this process, and the GPU box admits only a few processes on it at once: before
    starting ranks, wait until no process but this one holds the GPU any more (the ranks of the previous test may still be
    tearing their contexts down, and a launcher that merely imported torch holds /dev/kfd too)."""
    import time
    me = os.getpid()
    t0 = time.time()
    while True:
        others = []
        for p in os.listdir("/proc"):
            if not p.isdigit() or int(p) == me:
                continue
            try:
                for fd in os.listdir(f"/proc/{p}/fd"):
                    tgt = os.readlink(f"/proc/{p}/fd/{fd}")
                    if tgt.endswith("/kfd") or "renderD" in tgt:
                        others.append(int(p))
                        break
            except OSError:
                continue
        if not others or time.time() - t0 > timeout:
            return others
        time.sleep(0.25)
