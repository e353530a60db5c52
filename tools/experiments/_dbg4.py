import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims=[int(v) for v in sys.argv[1].split(",")]; n=int(sys.argv[2]); act=int(sys.argv[3]); lik=int(sys.argv[4]); prior=int(sys.argv[5])
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
if lik == o.LIK_BERNOULLI: theta = (theta * 0.3).astype(np.float32)
lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood, jit=True)
print(ch.kernel_name)
ch.set_data(X, Y)
prev = None
for rep in range(4):
    lp, g, st = ch.logp_grad(theta, eta)
    print("rep", rep, "lp", lp, lp64, "finite", bool(np.isfinite(g).all()), "same as before", None if prev is None else bool(np.array_equal(prev, g, equal_nan=True)))
    prev = g
    for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
        for nm, a, b in (("W", ow, ob), ("b", ob, ob + l.out_dim)):
            sc = max(np.abs(g64[a:b]).max(), 1e-3)
            e = np.abs(g[a:b] - g64[a:b]).max() / sc
            if not e <= 1e-4:
                print(f"   layer {li} {nm}: {e:.2e}")
                if nm == "W":
                    Wg = g[a:b].reshape(l.out_dim, l.in_dim); W0 = g64[a:b].reshape(l.out_dim, l.in_dim)
                    badm = ~(np.abs(Wg - W0) <= 1e-4 * sc); rows, cols = np.nonzero(badm)
                    print("     bad", int(badm.sum()), "rows", sorted(set((rows // 16).tolist())), "(tiles) cols", sorted(set((cols // 16).tolist())), "(tiles)")
ch.close()
