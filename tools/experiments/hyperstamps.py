"""dev: phase stamps of one leapfrog step of k_hyper (diagnostic build, see tools/tilestamps_run.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np
import tensorbnn_amd._native as nat
dbg = C.CDLL(os.path.join(os.path.dirname(nat.__file__), 'libtbnn_dbg.so'))
for name, res, args in nat.SYMBOLS:
    fn = getattr(dbg, name); fn.restype = res; fn.argtypes = args
nat.lib = dbg
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5, 50, 50, 50, 1], 4096)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
ch.logp_grad()
for _ in range(2): out = ch.hyper_step(1e-4, 100)
st = (C.c_uint64 * 64)()
dbg.tbnn_debug_tile_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
dbg.tbnn_debug_tile_stamps(ch._h, st)
t = np.array(list(st), dtype=np.float64)[40:46]
names = ["barrier A", "partials", "barrier B", "finish", "kick/drift"]
for i, nm in enumerate(names): print(f"{nm:12s} {t[i + 1] - t[i]:8.0f} cycles")
print("step total", t[5] - t[0], "cycles; device us per transition", out["device_us"])
