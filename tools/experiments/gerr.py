import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8"); os.environ.setdefault("TBNN_JIT", "0")
import numpy as np, tbnn_oracle as o
from tensorbnn_amd import _native as nat
for dims, n in (([5,50,50,50,1], 2000), ([5,50,50,50,1], 100000), ([6,51,51,1], 700)):
    spec, X, Y, theta, eta = o.synth_problem(dims, n)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = nat.Chain(layers, likelihood=spec.likelihood, jit=True)
    ch.set_data(X, Y)
    rng = np.random.default_rng(3)
    for trial in range(3):
        th = (theta * (1 + 0.05 * rng.standard_normal(theta.size))).astype(np.float32)
        lp, g, st = ch.logp_grad(th, eta)
        lp64, g64 = o.target_log_prob_and_grad(spec, th, eta, X, Y, np.float64)[:2]
        lp32, g32 = o.target_log_prob_and_grad(spec, th, eta, X, Y, np.float32)[:2]
        blocks = [(a, b) for l, (ow, ob) in zip(spec.layers, spec.offsets()) for a, b in ((ow, ob), (ob, ob + l.out_dim))]
        e = [np.abs(g[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for a, b in blocks]
        e32 = [np.abs(g32[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for a, b in blocks]
        print(ch.kernel_name, n, "logp rel", abs(lp - lp64) / abs(lp64), "grad err per tensor", " ".join(f"{x:.1e}" for x in e), "| fp32 oracle", " ".join(f"{x:.1e}" for x in e32))
    ch.close()
