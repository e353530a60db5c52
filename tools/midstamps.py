"""dev: shader-clock stamps of one tile of k_fwd_bwd_mid (build with TBNN_MID_FLAGS=-DMID_STAMPS): cycles per phase of
workgroup 0 / wave 0's second tile.  python tools/midstamps.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np
import tensorbnn_amd._native as nat
from tensorbnn_amd.workloads import synth_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
layers, lik, X, Y, th, eta = synth_problem([20, 100, 100, 2], n, likelihood=nat.LIK_BERNOULLI)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
print(ch.kernel_name)
for _ in range(3): ch.logp_grad()
out = (C.c_uint64 * 64)()
nat.lib.tbnn_debug_mid_stamps.argtypes = [C.POINTER(C.c_uint64)]
nat.lib.tbnn_debug_mid_stamps(out)
t = np.array(list(out), dtype=np.float64)
names = ["layer 0 fwd (35 MFMA)", "middle fwd (175)", "last layer + likelihood + delta_1 (VALU)", "delta chain (175) + relu'", "dW_1 (196)", "dW_0 (56)"]
for i, nm in enumerate(names):
    print(f"{nm:45s} {t[i + 1] - t[i]:8.0f} cycles")
print(f"{'tile':45s} {t[6] - t[0]:8.0f} cycles (637 MFMAs = {637 * 32})")
sub = lambda a, b: t[b] - t[a]
print("last layer: dots o0 %d | lane sum %d | act+lik %d | dots o1 %d | lane sum %d | act+lik %d | dW_L + delta_1 %d" % (
    sub(2, 16), sub(16, 17), sub(17, 18), sub(18, 19), sub(19, 20), sub(20, 21), sub(21, 3)))
print("delta chain: first-group requests + W^T MFMAs %d | relu' %d" % (sub(3, 25), sub(25, 4)))
