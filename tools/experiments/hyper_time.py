"""dev: what one hyper transition (L_h leapfrog steps in ONE k_hyper launch) costs next to a weight transition"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
for dims, n, lik, L in (([1, 10, 10, 1], 1000, o.LIK_GAUSSIAN, 100), ([5, 50, 50, 50, 1], 100000, o.LIK_GAUSSIAN, 50),
                        ([20, 100, 100, 2], 500000, o.LIK_BERNOULLI, 50), ([784, 20, 20, 1], 12000, o.LIK_BERNOULLI, 50)):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    ch.hmc_step(1e-7, 2); ch.hyper_step(1e-6, 4)
    for Lh in (10, 100):
        t = time.perf_counter()
        for _ in range(20): ch.hyper_step(1e-6, Lh)
        dt = (time.perf_counter() - t) / 20
        print(f"{dims} P={spec.n_params}: hyper transition L_h={Lh}: {dt * 1e6:.1f} us ({dt / Lh * 1e6:.2f} us per leapfrog step)")
    t = time.perf_counter()
    for _ in range(5): ch.hmc_step(1e-7, L)
    dt = (time.perf_counter() - t) / 5
    print(f"    weight transition L={L}: {dt * 1e6:.1f} us")
    ch.close()
