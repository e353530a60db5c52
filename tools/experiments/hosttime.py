"""dev: host-side cost of the per-epoch native calls (wall - device) at configs[1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5, 50, 50, 50, 1], 100000)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
for _ in range(3): ch.hmc_step(2e-5, 50); ch.hyper_step(1e-4, 100)
def tm(f, n=50):
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return 1e6 * (time.perf_counter() - t0) / n, r
w, out = tm(lambda: ch.hmc_step(2e-5, 50)); print(f"hmc_step   wall {w:8.1f} us  device {out['device_us']:8.1f} us  host {w - out['device_us']:6.1f} us")
w, out = tm(lambda: ch.hyper_step(1e-4, 100)); print(f"hyper_step wall {w:8.1f} us  device {out['device_us']:8.1f} us  host {w - out['device_us']:6.1f} us")
w, _ = tm(ch.get_state); print(f"get_state  wall {w:8.1f} us")
w, _ = tm(ch.get_hypers); print(f"get_hypers wall {w:8.1f} us")
