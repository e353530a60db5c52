"""dev: randomized parity sweep of run-time instantiated narrow kernels (fast3 / fast) against the fp64 oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
widths = [16, 17, 18, 32, 33, 34, 48, 49, 50, 64, 10, 24]
acts = [o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU]
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    nh = int(rng.integers(1, 4))
    dims = [int(rng.integers(1, 17))] + [int(rng.choice(widths)) for _ in range(nh)] + [int(rng.integers(1, 3))]
    act = int(rng.choice(acts)); prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    n = int(rng.integers(100, 3000))
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    try:
        ch = nat.Chain(layers, likelihood=spec.likelihood, kernel=nat.KERNEL_FAST, jit=True)
    except Exception as e:
        print(dims, "no narrow kernel:", str(e)[:60]); continue
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    e_lp = abs(lp - lp64) / max(abs(lp64), 1e-9)
    e_g = max(np.abs(g[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3)
              for l, (ow, ob) in zip(spec.layers, spec.offsets()) for a, b in ((ow, ob), (ob, ob + l.out_dim)))
    ok = e_lp <= 4e-6 and e_g <= 1e-4
    bad += not ok
    print(("ok  " if ok else "FAIL"), dims, "act", act, "prior", prior, "n", n, ch.kernel_name[:40], f"logp rel {e_lp:.1e} grad {e_g:.1e}")
    ch.close()
print("failures:", bad)
