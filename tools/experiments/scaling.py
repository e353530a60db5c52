import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 16384*8)
for k in (0, 1, 2, 4, 7, 8):
    n = max(16, 16384*k)
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X[:n], Y[:n]); ch.set_state(th); ch.set_hypers(eta)
    ch.hmc_run(1e-6, 10, 2)
    ch.set_profiling(1)
    o = ch.hmc_run(1e-6, 20, 3)
    print('tiles/wave', k, 'n', n, 'fwdbwd us', o[0]['fwdbwd_us'], 'epoch us/step', o[0]['device_us']/20)
    ch.close()
