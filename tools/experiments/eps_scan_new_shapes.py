import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, tbnn_oracle as o
from tensorbnn_amd import _native as nat
for dims, n, lik, xs, L, epss in (([784,20,20,1],12000,o.LIK_BERNOULLI,1/28.0,10,(8e-3,1.6e-2,3.2e-2,6.4e-2)), ([100,50,50,1],100000,o.LIK_GAUSSIAN,None,5,(2.4e-5,2.8e-5,3.2e-5,3.6e-5))):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    if xs: X = (np.abs(X) * xs).astype(np.float32)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    for eps in epss:
        ch = nat.Chain(layers, likelihood=spec.likelihood); ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        outs = ch.hmc_run(eps, L, 30)
        print(dims, eps, "accept", np.mean([x["accept_prob"] for x in outs]), flush=True)
        ch.close()
