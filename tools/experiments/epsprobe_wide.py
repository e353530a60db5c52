import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, numpy as np
sys.path.insert(0, "oracle")
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
for dims, n, lik in (([20,100,100,2], 500_000, o.LIK_BERNOULLI),):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = nat.Chain(layers, likelihood=spec.likelihood, kernel=nat.KERNEL_FAST)
    ch.set_data(X, Y)
    p0 = np.random.default_rng(3).standard_normal(spec.n_params).astype(np.float32)
    for eps in (4e-4, 2e-4, 1e-4, 5e-5, 2.5e-5, 1.25e-5):
        ch.set_state(theta); ch.set_hypers(eta)
        L = int(round(8e-4 / eps))
        out = ch.hmc_step(eps, L, p0=p0, log_u=1e30)
        print(dims, "eps", eps, "L", L, "lar", out["log_accept_ratio"], "dlogp", out["logp_new"] - out["logp_old"])
    ch.close()
