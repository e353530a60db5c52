import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: per-chunk shader-clock stamps of k_chain_wide (build with TBNN_EXTRA_FLAGS=-DWIDE_STAMPS)"""
import ctypes as C, numpy as np, sys
import tensorbnn_amd._native as nat
from tensorbnn_amd.workloads import synth_problem
case = sys.argv[1] if len(sys.argv) > 1 else "c4"
dims, n, lik = {"c4": ([10, 200, 200, 200, 1], 1_000_000, nat.LIK_GAUSSIAN), "c5": ([20, 100, 100, 2], 500_000, nat.LIK_BERNOULLI)}[case]
layers, lik, X, Y, th, eta = synth_problem(dims, n, likelihood=lik)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
for _ in range(3): ch.logp_grad()
out = (C.c_uint64 * 256)()
nat.lib.tbnn_debug_wide_stamps.argtypes = [C.POINTER(C.c_uint64)]
nat.lib.tbnn_debug_wide_stamps(out)
t = np.array(list(out), dtype=np.float64)
print("phases: L0 %d | fwd mid %d | last %d | bwd mid %d | dW0 %d | total %d" % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[5]-t[0]))
ends = [t[9 + 2 * c] for c in range(120) if t[9 + 2 * c] > 0]
print("chunk-to-chunk cycles (end of barrier to end of barrier; segment boundaries include the layer epilogues):")
print(" ".join("%d" % (b - a) for a, b in zip(ends[:-1], ends[1:])))
