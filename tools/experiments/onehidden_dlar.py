"""dev (GPU): is a log-accept-ratio difference of a free-running epoch the kernel's or the problem's?  One of tests/test_gpu_onehidden.py's cases epoch by
epoch (oracle set back on the device's state each epoch): device against the fp64 oracle AND the fp32 oracle against the fp64 oracle.
    python tools/experiments/onehidden_dlar.py wide_hidden_20_100_1 2e-4"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("TBNN_JIT", "1")
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as native
import test_gpu_onehidden as T
from test_gpu_freerun import draws, layers_of, SEED

name, eps = sys.argv[1], float(sys.argv[2])
spec, X, Y, theta, eta = T.problem(name)
ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True, seed=SEED, chain_id=2)
print(ch.kernel_name)
ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta); ch.set_epoch(0)
th = theta.astype(np.float64)
with np.errstate(all="ignore"):
    for ep in range(12):
        rec = ch.hmc_run(eps, 5, 1)[0]
        p0e, lu = draws(spec.n_params, 2, ep)
        r64 = o.weight_step(spec, th, eta, X, Y, eps, 5, p0e, lu, np.float64)
        r32 = o.weight_step(spec, th.astype(np.float32), eta, X, Y, eps, 5, p0e, lu, np.float32)
        tol = 2e-2 + 1e-4 * abs(r64.log_accept_ratio) + 1e-6 * abs(r64.logp_old)
        print(f"epoch {ep:2d}: lar fp64 {r64.log_accept_ratio:12.3f}  device - fp64 {rec['log_accept_ratio'] - r64.log_accept_ratio:9.3f}  fp32 oracle - fp64 "
              f"{r32.log_accept_ratio - r64.log_accept_ratio:9.3f}  (tol {tol:.3f}, logp {r64.logp_old:.4g})")
        th = ch.get_state().astype(np.float64)
ch.close()
