"""dev (GPU): which path the leapfrog steps of a transition take (per-step kernels or the trajectory kernel) for a few small networks, and the time per step"""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'oracle'))
os.environ["TBNN_JIT"] = "1"
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
for dims, n in (([1,10,10,1],1000),([1,100,1],1000),([1,100,1],11),([1,64,1],1000),([1,200,1],1000)):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_TANH, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    ch.hmc_run(1e-4, 50, 5)
    t = time.perf_counter(); ch.hmc_run(1e-4, 50, 40); dt = time.perf_counter() - t
    print(dims, n, ch.kernel_name, "| path:", ch.last_transition_path, "| %.2f us per leapfrog step" % (dt / 2000 * 1e6))
    ch.close()
