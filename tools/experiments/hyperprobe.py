import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([20,100,100,2], 500000, likelihood=nat.LIK_BERNOULLI)
for eh in (3e-5, 1e-5, 3e-6, 1e-6):
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    acc = []
    for e in range(12):
        ch.hmc_step(5e-5, 50)
        acc.append(ch.hyper_step(eh, 100)["accept_prob"])
    print("eps_h", eh, "hyper accept", np.round(acc, 2), "eta", np.round(ch.get_hypers()[:4], 3))
    ch.close()
