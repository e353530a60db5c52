"""dev: per-layer gradient error of the mid-width kernel (JIT) against the fp64 oracle: python tools/experiments/mid_debug.py 8,80,80,2 1200 [bern]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims = [int(v) for v in sys.argv[1].split(",")]
n = int(sys.argv[2])
bern = len(sys.argv) > 3 and sys.argv[3] == "bern"
act = {"relu": o.ACT_RELU, "tanh": o.ACT_TANH, "sigmoid": o.ACT_SIGMOID}[sys.argv[4]] if len(sys.argv) > 4 else o.ACT_RELU
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_GAUSSIAN if act == o.ACT_TANH else o.PRIOR_CAUCHY, o.LIK_BERNOULLI if bern else o.LIK_GAUSSIAN)
layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
ch = nat.Chain(layers, likelihood=spec.likelihood, kernel=nat.KERNEL_FAST, jit=os.environ.get("TBNN_JIT", "1") != "0")
print(ch.kernel_name)
ch.set_data(X, Y)
lp, g, st = ch.logp_grad(theta, eta)
lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
print("logp", lp, lp64)
for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
    dW = (g[ow:ob] - g64[ow:ob]).reshape(l.out_dim, l.in_dim)
    db = g[ob:ob + l.out_dim] - g64[ob:ob + l.out_dim]
    print(f"layer {li}: max|dW err| {np.abs(dW).max():.3e} (max |dW| {np.abs(g64[ow:ob]).max():.3e}), max|db err| {np.abs(db).max():.3e}")
    bad = np.argwhere(np.abs(dW) > 1e-3 * np.abs(g64[ow:ob]).max())
    if len(bad):
        print("   bad rows", sorted(set(bad[:, 0].tolist()))[:40], "bad cols", sorted(set(bad[:, 1].tolist()))[:40])
    badb = np.argwhere(np.abs(db) > 1e-3 * max(np.abs(g64[ob:ob + l.out_dim]).max(), 1e-6)).ravel()
    if len(badb):
        print("   bad bias", badb.tolist()[:40])
