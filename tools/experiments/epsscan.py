import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 100000)
for eps in (4e-5, 6e-5, 8e-5, 1e-4, 1.2e-4, 1.5e-4, 2e-4):
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    w = ch.hmc_run(2e-5, 50, 20)
    o = ch.hmc_run(eps, 50, 200)
    ap = np.array([x['accept_prob'] for x in o]); ac = np.array([x['accepted'] for x in o])
    print('eps', eps, 'timed acc_prob', round(float(ap.mean()),3), 'accepted frac', round(float(ac.mean()),3), 'q1..q4', [round(float(ap[i*50:(i+1)*50].mean()),2) for i in range(4)], 'logp end', round(o[-1]['logp_old'],1))
    ch.close()
