"""Build libtbnn.so in-tree with hipcc for gfx950 (`python -m tensorbnn_amd.build`)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "tbnn_api.hip"), os.path.join(HERE, "csrc", "adapter.cpp")]
OUT = os.path.join(HERE, "libtbnn.so")


def _deps():
    d = list(SRC)
    cs = os.path.join(HERE, "csrc")
    d += [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(".hpp")]
    d.append(os.path.join(HERE, "..", "include", "tbnn.h"))
    return d


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in _deps()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-value", "-Wno-unused-result", "-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
