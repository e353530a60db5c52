"""dev: per-block gradient errors of a narrow-family shape against the fp64 oracle (which part of dW is off: full tiles, fringe rows,
N-fringe columns, bias):  python tools/experiments/nfdbg.py 7,17,33,2 700 [act lik prior grid]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,50,50,50,1").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
act = int(sys.argv[3]) if len(sys.argv) > 3 else o.ACT_RELU
lik = int(sys.argv[4]) if len(sys.argv) > 4 else o.LIK_GAUSSIAN
prior = int(sys.argv[5]) if len(sys.argv) > 5 else o.PRIOR_CAUCHY
if len(sys.argv) > 6: os.environ["TBNN_FAST_GRID"] = sys.argv[6]
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood, jit=True)
print(ch.kernel_name)
ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
lp, g, st = ch.logp_grad(theta, eta)
lp0, g0 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
print(lp, lp0)
for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
    W = g[ow:ob].reshape(l.out_dim, l.in_dim); W0 = g0[ow:ob].reshape(l.out_dim, l.in_dim)
    b = g[ob:ob + l.out_dim]; b0 = g0[ob:ob + l.out_dim]
    e = np.abs(W - W0) / np.abs(W0).max(); eb = np.abs(b - b0) / np.abs(b0).max()
    fu, fc = 16 * (l.out_dim // 16), 16 * (l.in_dim // 16)
    f = lambda a: "%.1e" % a.max() if a.size else "-"
    print(f"layer {li} [{l.out_dim} x {l.in_dim}]: full x full {f(e[:fu, :fc])} | full units x fringe cols {f(e[:fu, fc:])} | fringe units x full cols {f(e[fu:, :fc])} | "
          f"corner {f(e[fu:, fc:])} | bias full {f(eb[:fu])} fringe {f(eb[fu:])}")
