"""dev: random long-fan-in architectures through the run-time instantiation of the tall family (forced with TBNN_JIT_SKIP), random row counts
and forced group sizes, value / gradient against the fp64 oracle:  python tools/experiments/tall_fuzz.py [n_shapes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
os.environ["TBNN_JIT_SKIP"] = "fast3,fast,mid,wide"
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat, jit
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for k in range(N):
    d_in = int(rng.choice([rng.integers(33, 129), rng.integers(129, 600), rng.integers(600, 1300)]))
    nh = int(rng.integers(1, 4))
    hidden = [int(rng.integers(3, 65)) for _ in range(nh)]
    d_out = int(rng.integers(1, 3))
    dims = [d_in] + hidden + [d_out]
    if not jit.tall_fits(dims):
        print(dims, "does not fit the tall family: skipped"); continue
    act = int(rng.choice([o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID]))
    lik = int(rng.choice([o.LIK_GAUSSIAN, o.LIK_BERNOULLI]))
    prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 20000)]))
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    X = (np.abs(X) / np.sqrt(d_in)).astype(np.float32)
    if lik == o.LIK_BERNOULLI:
        Y = (rng.random(Y.shape) < 0.5).astype(np.float32)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    for G in ("auto", int(rng.integers(1, 5))):
        if G == "auto": os.environ.pop("TBNN_TALL_G", None)
        else: os.environ["TBNN_TALL_G"] = str(G)
        t = time.time()
        try:
            ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True)
        except Exception as e:
            print(dims, "create failed:", str(e)[:100]); bad += 1; break
        name = ch.kernel_name
        ch.set_data(X, Y)
        lp, g, st = ch.logp_grad(theta, eta)
        f = ch.forward_many(theta[None, :], X=X[: min(n, 500)])[0]
        f64 = o.forward(spec, theta, X[: min(n, 500)], np.float64)
        ch.close()
        e_lp = abs(lp - lp64) / max(abs(lp64), 1.0)
        e_g = max(np.abs(g[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for l, (ow, ob) in zip(spec.layers, spec.offsets()) for a, b in ((ow, ob), (ob, ob + l.out_dim)))
        e_f = float(np.abs(f - f64).max())
        ok = e_lp <= 4e-6 and e_g <= 1e-4 and e_f <= 1e-4          # (a shape whose instantiation spills falls to the layered family: still checked)
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {dims} n={n} act={act} lik={lik} prior={prior} G={G}: {name}; logp {e_lp:.1e} grad {e_g:.1e} forward {e_f:.1e} ({time.time() - t:.0f} s)", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
