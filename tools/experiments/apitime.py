"""dev: throughput of the Python API (network.train) at configs[1] against the native bench loop: per-epoch cost of
the hyper transition, the state read-back and the host loop"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensorbnn_amd.network import network
from tensorbnn_amd.layer import DenseLayer
from tensorbnn_amd.activationFunctions import Relu
from tensorbnn_amd.likelihood import GaussianLikelihood
from tensorbnn_amd.workloads import synth_problem

dims, n, L = [5, 50, 50, 50, 1], 100000, 50
layers, lik, X, Y, th, eta = synth_problem(dims, n)
for hypers in (False, True):
    net = network(np.float32, dims[0], X, Y, X[:1000], Y[:1000])
    off = 0
    for i in range(len(dims) - 1):
        w = th[off:off + dims[i] * dims[i + 1]].reshape(dims[i + 1], dims[i]); off += w.size
        b = th[off:off + dims[i + 1]].reshape(dims[i + 1], 1); off += b.size
        net.add(DenseLayer(dims[i], dims[i + 1], weights=w, biases=b))
        if i + 2 < len(dims):
            net.add(Relu())
    net.setupMCMC(stepSizeStart=5e-5, leapfrogStart=L, hyperStepSize=1e-4, hyperLeapfrog=100, burnin=10, adapt=False)
    net._ensure_chain(GaussianLikelihood(sd=0.1))            # chain creation / data staging outside the timed call
    t0 = time.perf_counter()
    rec = net.train(40, 1, GaussianLikelihood(sd=0.1), adjustHypers=hypers, verbose=False)
    dt = time.perf_counter() - t0
    dev = np.mean([r["main"]["device_us"] for r in rec])
    hy = np.mean([r["hyper"]["device_us"] for r in rec]) if hypers else 0.0
    print(f"adjustHypers={hypers}: {1e3 * dt / 40:.3f} ms/epoch wall ({40 * L / dt:.0f} leapfrog steps/s); weight transition {dev / 1e3:.3f} ms, "
          f"hyper transition {hy / 1e3:.3f} ms on the device; accept {np.mean([r['main']['accept_prob'] for r in rec]):.2f}")
