"""dev: network.train as a user runs it at configs[1] -- adapter on, hypers on, samples written, metrics displayed --
against the device time of the transitions (host loop overhead = the difference)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tensorbnn_amd.network import network
from tensorbnn_amd.layer import DenseLayer
from tensorbnn_amd.activationFunctions import Relu
from tensorbnn_amd.likelihood import GaussianLikelihood
from tensorbnn_amd.metrics import SquaredError, PercentError
from tensorbnn_amd.workloads import synth_problem

dims, n = [5, 50, 50, 50, 1], 100000
layers, lik, X, Y, th, eta = synth_problem(dims, n)
net = network(np.float32, dims[0], X, Y, X[:5000], Y[:5000])
off = 0
for i in range(len(dims) - 1):
    w = th[off:off + dims[i] * dims[i + 1]].reshape(dims[i + 1], dims[i]); off += w.size
    b = th[off:off + dims[i + 1]].reshape(dims[i + 1], 1); off += b.size
    net.add(DenseLayer(dims[i], dims[i + 1], weights=w, biases=b))
    if i + 2 < len(dims):
        net.add(Relu())
net.setupMCMC(stepSizeStart=2e-5, stepSizeMin=1e-5, stepSizeMax=1e-4, stepSizeOptions=40, leapfrogStart=50, leapfogMin=20,
              leapFrogMax=100, leapfrogIncrement=1, hyperStepSize=1e-4, hyperLeapfrog=100, burnin=40, averagingSteps=5)
net._ensure_chain(GaussianLikelihood(sd=0.1))
os.chdir(tempfile.mkdtemp())
E = 120
t0 = time.perf_counter()
rec = net.train(E, 5, GaussianLikelihood(sd=0.1), metricList=[SquaredError(), PercentError()], adjustHypers=True,
                folderName="run", networksPerFile=50, displaySkip=40, verbose=False)
dt = time.perf_counter() - t0
steps = sum(r["L"] for r in rec)
dev = sum(r["main"]["device_us"] + r["hyper"]["device_us"] for r in rec) * 1e-6
print(f"{E} epochs, {steps} leapfrog steps: wall {dt:.3f} s ({steps / dt:.0f} steps/s), transitions on the device {dev:.3f} s "
      f"({100 * dev / dt:.0f} %), host loop {1e3 * (dt - dev) / E:.3f} ms/epoch; mean accept {np.mean([r['main']['accept_prob'] for r in rec]):.2f}")
