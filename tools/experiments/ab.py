import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, os, sys, numpy as np
import tensorbnn_amd._native as nat
from tensorbnn_amd.workloads import synth_problem
layers, lik, X, Y, th, eta = synth_problem([5,50,50,50,1], 100000)
import glob
libs = sorted(glob.glob(os.path.join(os.path.dirname(nat.__file__), 'libtbnn*.so')))
for lp in libs:
    lib = C.CDLL(lp)
    for name, res, args in nat.SYMBOLS:
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    nat.lib = lib
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    ch.hmc_run(2e-5, 50, 5)
    ch.set_profiling(5)
    o = ch.hmc_run(2e-5, 50, 20)
    lp_, g_, _ = ch.logp_grad(th, eta)
    print(os.path.basename(lp), ch.kernel_name, 'fwdbwd us', round(o[0]['fwdbwd_us'],2), 'epoch us/step', round(o[0]['device_us']/50,2), 'logp', lp_)
    ch.close()
