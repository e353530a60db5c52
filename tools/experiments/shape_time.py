"""dev: which kernel serves a shape and what one gradient costs:  python tools/experiments/shape_time.py 784,20,20,1 12000 [bern|gauss] [relu|tanh|sigmoid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims = [int(x) for x in sys.argv[1].split(",")]
n = int(sys.argv[2])
lik = o.LIK_BERNOULLI if len(sys.argv) > 3 and sys.argv[3] == "bern" else o.LIK_GAUSSIAN
act = {"relu": o.ACT_RELU, "tanh": o.ACT_TANH, "sigmoid": o.ACT_SIGMOID}[sys.argv[4] if len(sys.argv) > 4 else "relu"]
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, lik)
layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd)
print("kernel:", ch.kernel_name, flush=True)
ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
lp, g, st = ch.logp_grad(theta, eta)
if n * sum(a * b for a, b in zip(dims[:-1], dims[1:])) <= 4e8:
    lp0, g0 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    print(f"logp {lp:.6f} vs fp64 {lp0:.6f} (rel {abs(lp - lp0) / abs(lp0):.2e}); grad err {np.abs(g - g0).max() / np.abs(g0).max():.2e}")
for L in (20,):
    ch.hmc_step(1e-6, 3)
    t = time.perf_counter(); out = ch.hmc_step(1e-6, L); dt = time.perf_counter() - t
    flop = 2 * n * (3 * sum(a * b for a, b in zip(dims[:-1], dims[1:])) - dims[0] * dims[1])
    print(f"{dt / L * 1e6:.1f} us per leapfrog step = {flop / (dt / L) / 1e12:.2f} TFLOP/s ({flop / (dt / L) / 157.3e12:.4f} of peak)")
ch.close()
