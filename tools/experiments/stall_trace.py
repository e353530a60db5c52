"""dev: per-epoch wall time of chains created one after the other (with / without closing the previous one): finds single long stalls.
Round 4: the 77-80 ms stalls it found were the container being throttled by its CPU quota after a multi-threaded NumPy call (NOTES.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import WORKLOADS, burned_state, synth_problem
probs = {}
def mk(name):
    wl = WORKLOADS[name]
    if name not in probs:
        probs[name] = synth_problem(wl["dims"], wl["n"], prior=wl["prior"], likelihood=wl["lik"], x_scale=wl.get("x_scale"))
    layers, lik, X, Y, theta0, eta0 = probs[name]
    b = burned_state(name, os.path.join(ROOT, "tests", "golden"))
    t = time.perf_counter()
    ch = nat.Chain(layers, likelihood=lik)
    ch.set_data(X, Y); ch.set_state(b["theta"].astype(np.float32)); ch.set_hypers(b["eta"].astype(np.float32))
    return ch, float(b["eps"]), wl["L"], time.perf_counter() - t
def trace(ch, eps, L, n, label):
    ts = []
    t00 = time.perf_counter()
    for i in range(n):
        t = time.perf_counter(); ch.hmc_run(eps, L, 1); ts.append((time.perf_counter() - t) * 1e3)
    med = float(np.median(ts))
    st = [(i, round(x, 1), round((sum(ts[:i])), 1)) for i, x in enumerate(ts) if x > 3 * med]
    print(f"{label}: median {med:.2f} ms, total {sum(ts):.0f} ms; stalls (epoch, ms, ms since first launch): {st}", flush=True)
a, e, L, tc = mk("mn"); print(f"create {tc*1e3:.0f} ms"); trace(a, e, L, 120, "A fresh")
b, e2, L2, tc = mk("mn"); print(f"create {tc*1e3:.0f} ms"); trace(b, e2, L2, 120, "B (A alive)")
a.close(); b.close()
c, e3, L3, tc = mk("mn"); print(f"create {tc*1e3:.0f} ms"); trace(c, e3, L3, 120, "C (after closing A, B)")
trace(c, e3, L3, 120, "C again")
c.close()
d, e4, L4, tc = mk("c1"); print(f"create {tc*1e3:.0f} ms"); trace(d, e4, L4, 300, "D = c1 (after closing C)")
d.close()
f, e5, L5, tc = mk("c2"); print(f"create {tc*1e3:.0f} ms"); trace(f, e5, L5, 60, "F = c2 (after closing D)")
