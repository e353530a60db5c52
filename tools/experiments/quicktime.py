import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time, numpy as np
sys.path.insert(0, 'oracle')
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 100000)
layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
for kern in (nat.KERNEL_GENERIC, nat.KERNEL_AUTO):
    ch = nat.Chain(layers, likelihood=spec.likelihood, kernel=kern)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    t0=time.time(); lp, g, st = ch.logp_grad(); t1=time.time()
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    print(ch.kernel_name, 'logp', lp, lp64, 'rel', abs(lp-lp64)/abs(lp64), 'grad relinf', np.abs(g-g64).max()/np.abs(g64).max())
    for eps in (1e-5, 2e-5, 5e-5):
        ch.set_state(theta)
        outs = ch.hmc_run(eps, 50, 10)
        print(' eps', eps, 'acc', np.mean([x['accept_prob'] for x in outs]), 'us/epoch', outs[0]['device_us'], 'steps/s', 50/(outs[0]['device_us']*1e-6))
    ch.set_profiling(True)
    out = ch.hmc_step(2e-5, 50)
    print(' profiled: epoch us', out['device_us'], 'fwdbwd us total', out['fwdbwd_us'], 'per launch', out['fwdbwd_us']/50)
    ch.close()
