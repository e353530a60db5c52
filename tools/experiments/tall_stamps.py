"""dev: prologue / group loop / epilogue of k_fwd_bwd_tall in shader-clock cycles (diagnostic build):
TBNN_BUILD_TAG=stamps TBNN_TALL_FLAGS=-DTBNN_TILE_STAMPS python -m tensorbnn_amd.build
TBNN_LIB=$PWD/tensorbnn_amd/libtbnn_stamps.so python tools/experiments/tall_stamps.py 784,20,20,1 12000 bern"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims = [int(x) for x in sys.argv[1].split(",")]; n = int(sys.argv[2])
lik = o.LIK_BERNOULLI if len(sys.argv) > 3 and sys.argv[3] == "bern" else o.LIK_GAUSSIAN
spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood)
print(ch.kernel_name)
ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
for _ in range(5): ch.logp_grad(theta, eta)
lib = nat.lib
buf = (ctypes.c_ulonglong * 64)()
fn = lib.tbnn_tall_debug_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(buf) == 0
v = list(buf)
print("launch: prologue", v[17] - v[16], " group loop", v[18] - v[17], " dW_0 epilogue", v[19] - v[18], " rest of epilogue", v[20] - v[19], " total", v[20] - v[16])
print("first group: phase A", v[21] - v[17], " wait at barrier 1", v[22] - v[21], " phase B", v[23] - v[22], " wait at barrier 2", v[24] - v[23],
      " phase C (to the loop's end when the workgroup has one group)", v[18] - v[24])
