import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: time the wide path at full size (C4 / C5) -- uses workloads.synth_problem (no oracle), rocprof friendly"""
import sys, time
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem

case = sys.argv[1] if len(sys.argv) > 1 else "c4"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dims, n, lik = {"c4": ([10, 200, 200, 200, 1], 1_000_000, nat.LIK_GAUSSIAN),
                "c5": ([20, 100, 100, 2], 500_000, nat.LIK_BERNOULLI)}[case]
if len(sys.argv) > 3:
    n = int(sys.argv[3])
layers, lik, X, Y, theta0, eta0 = synth_problem(dims, n, likelihood=lik)
S = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
flops = 2.0 * n * (3 * S - dims[0] * dims[1])
ch = nat.Chain(layers, likelihood=lik, kernel=nat.KERNEL_AUTO)
ch.set_data(X, Y); ch.set_state(theta0); ch.set_hypers(eta0)
print(ch.kernel_name, "P", ch.P, "n", n, "GFLOP/grad", flops / 1e9)
eps = 1e-6
ch.hmc_run(eps, 2, 1)
ch.set_profiling(1)
t0 = time.time()
outs = ch.hmc_run(eps, L, 2)
dt = time.time() - t0
us = outs[0]["fwdbwd_us"]
print(f"fwd+bwd {us:.1f} us/launch -> {flops / us / 1e6:.2f} TFLOP/s = {flops / us / 1e6 / 157.3:.3f} of peak; "
      f"epoch {outs[0]['device_us']:.0f} us, {L / (outs[0]['device_us'] * 1e-6):.1f} leapfrog/s; accept {outs[0]['accept_prob']:.3f}")
ch.close()
