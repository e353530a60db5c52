"""dev: the narrow chain group of tests/test_gpu_freerun.py, per epoch: device log accept ratio against the fp64 and the fp32 oracle (both on the
oracle chain's state) -- tells an ill-conditioned epoch (the fp32 oracle is off by as much) from a kernel defect:
  python tools/experiments/freerun_dlar.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
os.environ.setdefault("TBNN_JIT", "0")
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
import test_gpu_freerun as T
dims, n, act, prior, lik, eps, kname = T.GROUPS["narrow"]
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
C, c0, L, BURN, E1 = 4, 5, 5, 60, 14
rng = np.random.default_rng(12)
thetas = (theta[None, :] * (1.0 + 0.05 * rng.standard_normal((C, theta.size)))).astype(np.float32)
grp = nat.ChainGroup(T.layers_of(spec), C, likelihood=spec.likelihood, seed=T.SEED, chain_id=c0, jit=False)
grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(eta)
grp.hmc_run(eps, L, BURN)
start, eta0 = grp.get_state(), grp.get_hypers()
r1 = grp.hmc_run(eps, L, E1)
grp.close()
with np.errstate(all="ignore"):
    for c in (1, 3):
        th, et = start[c].astype(np.float64), eta0[c].astype(np.float64)
        for k in range(E1):
            p0, lu = T.draws(spec.n_params, c0 + c, BURN + k)
            ref = o.weight_step(spec, th, et, X, Y, eps, L, p0, lu, np.float64)
            r32 = o.weight_step(spec, th.astype(np.float32), et.astype(np.float32), X, Y, eps, L, p0, lu, np.float32)
            tol = T.lar_tol(ref.log_accept_ratio, ref.logp_old)
            print(f"chain {c} epoch {k}: device {r1[c][k]['log_accept_ratio']:+.5f} fp64 {ref.log_accept_ratio:+.5f} fp32-oracle {r32.log_accept_ratio:+.5f} "
                  f"|dev-64|/tol {abs(r1[c][k]['log_accept_ratio'] - ref.log_accept_ratio) / tol:.3f} |32-64|/tol {abs(r32.log_accept_ratio - ref.log_accept_ratio) / tol:.3f}")
            dec = lu < ref.log_accept_ratio if abs(lu - ref.log_accept_ratio) >= T.MARGIN else bool(r1[c][k]["accepted"])
            if dec:
                th = ref.theta_proposed.astype(np.float64)
