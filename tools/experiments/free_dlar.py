"""dev: per-epoch |dlar| / band of a free-running chain against oracle/c (TBNN_LIB picks the library):
python tools/experiments/free_dlar.py 784,20,20,1 12000 bern 0.0357 5e-3 30 10 [burn]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import tbnn_oracle as o, c_oracle
from tensorbnn_amd import _native as nat
from test_gpu_fullsize import away_from, lar_tol
dims = [int(x) for x in sys.argv[1].split(",")]; n = int(sys.argv[2])
lik = o.LIK_BERNOULLI if sys.argv[3] == "bern" else o.LIK_GAUSSIAN
xs, eps, epochs, L = float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
burn = int(sys.argv[8]) if len(sys.argv) > 8 else 0
spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
if xs: X = (np.abs(X) * xs).astype(np.float32)
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood)
print(ch.kernel_name)
ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
if burn: ch.hmc_run(eps, 20, burn); theta = ch.get_state()
co = c_oracle.COracle(spec, X, Y); rng = np.random.default_rng(2025); th_c = theta.copy()
for ep in range(epochs):
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    q_c, lar_c, lp0_c, _ = co.hmc_propose(th_c, eta, eps, L, p0)
    lu = away_from(rng, lar_c)
    if lu < lar_c: th_c = q_c
    out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
    d = np.abs(ch.get_state() - th_c).max() / np.abs(th_c).max()
    print(f"{ep:3d} lar {out['log_accept_ratio']:+.4f} vs {lar_c:+.4f}  |d|/band {abs(out['log_accept_ratio'] - lar_c) / lar_tol(lar_c, lp0_c):.3f}  logp0 {lp0_c:.1f}  acc {int(out['accepted'])}/{int(lu < lar_c)}  state dist {d:.1e}")
