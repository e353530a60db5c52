"""dev: leapfrog steps/s of the plumbing-size configs (BASELINE configs[0]: launch-latency-bound)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
for dims, n, L in (([1, 10, 10, 1], 1000, 100), ([1, 10, 10, 10, 1], 1000, 100), ([5, 50, 50, 50, 1], 2000, 50)):
    layers, lik, X, Y, th, eta = synth_problem(dims, n)
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    ch.hmc_run(1e-4, L, 5)
    t0 = time.perf_counter(); outs = ch.hmc_run(1e-4, L, 40); t1 = time.perf_counter()
    dev = np.mean([o["device_us"] for o in outs])
    print(f"{dims} n={n} L={L} kernel={ch.kernel_name}: wall {1e6 * (t1 - t0) / 40 / L:.2f} us/step, device {dev / L:.2f} us/step, "
          f"{40 * L / (t1 - t0):.0f} steps/s, accept {np.mean([o['accept_prob'] for o in outs]):.2f}")
    ch.close()
