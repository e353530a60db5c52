"""dev (GPU): the mid-width kernel beyond its fan-in limit (MID_MAX_FANIN = 128; -DMID_MAX_FANIN=256 through TBNN_JIT_FLAGS, libraries compiled
here first) against the tall kernel:  python tools/experiments/mid_fanin.py 200,32,32,1 100000"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dims, n = sys.argv[1], sys.argv[2]
code = ("import sys; sys.path.insert(0, %r); from tensorbnn_amd import jit; jit.MID_MAX_FANIN = 256; sys.argv = ['shape_time.py', %r, %r, 'gauss'];"
        "__file__ = %r; exec(open(__file__).read())" % (root, dims, n, os.path.join(root, "tools", "experiments", "shape_time.py")))
for name, env in (("mid (fan-in limit 256)", {"TBNN_JIT_FLAGS": "-DMID_MAX_FANIN=256", "TBNN_JIT_SKIP": "fast3,fast,tall,wide"}),
                  ("tall", {"TBNN_JIT_SKIP": "fast3,fast,mid,wide"})):
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True)
    out = r.stdout if r.returncode == 0 else r.stdout + r.stderr[-400:]
    print(name, "|", " ".join(l for l in out.splitlines() if l.startswith(("kernel", "logp")) or "us per" in l))
