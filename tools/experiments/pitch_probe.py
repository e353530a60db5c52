"""round 2: what would conflict-free operand reads buy?  A 2-hidden-layer shape (fits LDS with either padding) compiled at run
time with the image paddings 4 (product: every b128 operand read 2-way bank-conflicted on gfx950) and 8 (conflict-free):
TBNN_JIT_FLAGS="-DTBNN_WPAD=8 -DTBNN_PPAD=8" python tools/experiments/pitch_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
dims = [5, 50, 50, 1]
layers, lik, X, Y, th, eta = synth_problem(dims, 100_000)
ch = nat.Chain(layers, likelihood=lik, kernel=nat.KERNEL_FAST, jit=True)
ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
ch.hmc_run(1e-5, 20, 2)
ch.set_profiling(5)
outs = ch.hmc_run(1e-5, 50, 6)
print(os.environ.get("TBNN_JIT_FLAGS", "(default)"), ch.kernel_name, "fwd+bwd us:", round(float(np.mean([o["fwdbwd_us"] for o in outs])), 2),
      "logp", ch.logp_grad()[0])
ch.close()
