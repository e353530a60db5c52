import sys, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 100000)
for eps in (1e-5, 1.5e-5, 2e-5, 2.5e-5, 3e-5, 4e-5):
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    w = ch.hmc_run(eps, 50, 20)
    o = ch.hmc_run(eps, 50, 200)
    ap = np.array([x['accept_prob'] for x in o])
    print('eps', eps, 'warm acc', np.mean([x['accept_prob'] for x in w]), 'timed acc', ap.mean(), 'first50', ap[:50].mean(), 'last50', ap[-50:].mean(), 'logp end', o[-1]['logp_old'], 'us/epoch', o[0]['device_us'])
    ch.close()
