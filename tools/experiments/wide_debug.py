import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: wide path vs generic kernel vs oracle on a small problem"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
import tbnn_oracle as o
from tensorbnn_amd import _native as nat

case = sys.argv[1] if len(sys.argv) > 1 else "t1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 517
cfg = {"t1": ([3, 20, 36, 2], o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
       "t2": ([20, 32, 16, 48, 2], o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
       "c5": ([20, 100, 100, 2], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
       "c4": ([10, 200, 200, 200, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)}[case]
spec, X, Y, theta, eta = o.synth_problem(cfg[0], n, cfg[1], cfg[2], cfg[3])
layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
res = {}
for name, k in (("generic", nat.KERNEL_GENERIC), ("wide", nat.KERNEL_FAST)):
    ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=k)
    ch.set_data(X, Y)
    lp, g, stat = ch.logp_grad(theta, eta)
    print(name, ch.kernel_name, "logp", lp, "stat", stat)
    res[name] = (lp, g)
    ch.close()
lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
print("oracle logp", lp64)
for name in res:
    g = res[name][1]
    for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
        for nm, a, b in (("W", ow, ob), ("b", ob, ob + l.out_dim)):
            ref = g64[a:b]
            err = np.abs(g[a:b] - ref).max()
            print(f"  {name} layer {li} {nm}: err {err:.3e} of max {np.abs(ref).max():.3e}")
