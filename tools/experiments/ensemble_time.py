import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: time tbnn_forward_many (ensemble prediction) for the configs[1] shape: m networks x n rows"""
import sys, time
import numpy as np
import tensorbnn_amd._native as nat
from tensorbnn_amd.workloads import synth_problem
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 20000)
layers, lik, X, Y, th, eta = synth_problem([5, 50, 50, 50, 1], n)
ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_validation(X, Y); ch.set_state(th)
rng = np.random.default_rng(0)
thetas = (th[None, :] * (1 + 0.1 * rng.standard_normal((m, th.size)))).astype(np.float32)
ch.forward_many(thetas[:4], which=1)
t0 = time.perf_counter(); out = ch.forward_many(thetas, which=1); t1 = time.perf_counter()
flop = 2.0 * n * 5300 * m
print(f"batched: {m} nets x {n} rows in {1e3 * (t1 - t0):.1f} ms (incl. {out.nbytes / 1e6:.0f} MB D2H) = {flop / (t1 - t0) / 1e12:.2f} TFLOP/s")
t0 = time.perf_counter()
for i in range(min(m, 50)): ch.predict(1, thetas[i])
t1 = time.perf_counter()
print(f"one call per network (tbnn_predict): {1e3 * (t1 - t0) / min(m, 50):.3f} ms per network")
