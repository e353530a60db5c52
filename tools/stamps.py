"""dev: 100 MHz wall-clock stamps of workgroup 0 inside one fused launch (tbnn_debug_stamps): prologue / tile loop /
cooperative tail / epilogue of k_fwd_bwd_fast3 at configs[1]'s size"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5, 50, 50, 50, 1], 114688)
for n in (16384, 98304, 100000, 114688):
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X[:n], Y[:n]); ch.set_state(th); ch.set_hypers(eta)
    out = (C.c_uint64 * 16)()
    nat.lib.tbnn_debug_stamps(ch._h, out)
    t = np.array(list(out)[:8], dtype=np.float64); d = (t - t[0]) * 0.01
    c = np.array(list(out)[8:16], dtype=np.float64)
    print(f"n {n}: prologue {d[1]:.2f} us, first tile end {d[2]:.2f}, tile loop end {d[3]:.2f}, cooperative tail end {d[5]:.2f}, "
          f"staged {d[6]:.2f}, tile slabs out {d[7]:.2f}, end {d[4]:.2f}; shader clock over the launch {(c[4] - c[0]) / (d[4] + 1e-9):.0f} MHz")
    ph = [("prologue", 0, 1), ("first tile", 1, 2), ("other tiles", 2, 3), ("coop", 3, 5), ("epilogue", 5, 4)]
    print("   shader clock per phase (MHz):", ", ".join(f"{nm} {(c[b] - c[a]) / max(d[b] - d[a], 1e-9):.0f}" for nm, a, b in ph),
          "| shader cycles:", ", ".join(f"{nm} {c[b] - c[a]:.0f}" for nm, a, b in ph))
    ch.close()
