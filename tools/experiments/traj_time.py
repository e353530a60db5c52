"""dev: leapfrog steps/s of configs[0]'s network over the row count, trajectory kernel (kernels_traj.hpp) against the two-kernel step:
  TBNN_TRAJ=1|0 python tools/experiments/traj_time.py [chains] [dims, default 1,10,10,1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
DIMS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 10, 10, 1]
for n in (64, 256, 512, 1000, 2000):
    spec, X, Y, theta, eta = o.synth_problem(DIMS, n, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_FIXED_GAUSSIAN if hasattr(o, "LIK_FIXED_GAUSSIAN") else o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    if C == 1:
        ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=0, jit=True)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    else:
        ch = nat.ChainGroup(layers, C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=0, jit=True)
        ch.set_data(X, Y); ch.set_state(np.tile(theta, (C, 1))); ch.set_hypers(np.tile(eta, (C, 1)))
    L, E = 100, 40
    ch.hmc_run(1e-4, L, 5)
    t = time.time(); ch.hmc_run(1e-4, L, E); dt = time.time() - t
    print(f"TBNN_TRAJ={os.environ.get('TBNN_TRAJ', '1')} chains {C} rows {n}: {C * L * E / dt / 1e3:.1f} k leapfrog steps/s ({dt / (L * E) * 1e6:.2f} us per step)", flush=True)
    ch.close()
