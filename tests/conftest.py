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


def pytest_collection_finish(session):
    """A GPU run on a tree that arrives WITHOUT its ahead-of-time kernel cache (tensorbnn_amd/_jit, which __graft_entry__.build() fills and which travels
    with the tree like libtbnn.so): compile the suite's run-time instantiations now, side by side on the box's CPUs (~2 min for ~140 libraries), instead
    of one by one inside the tests (~20 min).  With the cache in place every job is a cache hit (a second or two).  Before any test has touched the GPU:
    the workers are forked and only drive compiler subprocesses.  TBNN_PREBUILD_JIT=0 skips it."""
    if getattr(session.config.option, "collectonly", False) or os.environ.get("TBNN_PREBUILD_JIT", "1") == "0":
        return
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    try:
        import __graft_entry__ as ge
        ge.prebuild_jit()
    except Exception as e:          # never fails the run: the tests compile what is missing
        print("run-time kernel libraries not prebuilt:", e)


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
