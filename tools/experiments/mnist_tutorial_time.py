"""dev: wall time of the reference's MNIST tutorial run (docs/ClassificationExample.md:103-173) through the drop-in Python API:
784 -> 20 -> 20 -> 1, BernoulliLikelihood, the tutorial's setupMCMC arguments (eps 1e-3 in [5e-4, 2e-3] x 100, L 500 in
[100, 2000], hyper step 1e-5 x 30, burn-in 50, averaging 2), on synthetic 'digit' rows in [0, 1] (12,000 train / 2,000
validation: the size of MNIST's 3-vs-8 subset).  Relu instead of the tutorial's SquarePrelu.
  python tools/experiments/mnist_tutorial_time.py [epochs=100]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tensorbnn_amd.activationFunctions import Relu, Sigmoid
from tensorbnn_amd.layer import DenseLayer
from tensorbnn_amd.likelihood import BernoulliLikelihood
from tensorbnn_amd.metrics import Accuracy
from tensorbnn_amd.network import network

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(11)
proto = rng.random((2, 784)).astype(np.float32)
lab = (rng.random(14000) < 0.5).astype(np.float32)
X = np.clip(proto[lab.astype(int)] * 0.5 + 0.5 * rng.random((14000, 784)), 0, 1).astype(np.float32)
os.chdir(tempfile.mkdtemp())
net = network(np.float32, 784, X[:12000], lab[:12000, None], X[12000:], lab[12000:, None])
net.add(DenseLayer(784, 20, seed=0)); net.add(Relu())
net.add(DenseLayer(20, 20, seed=1000)); net.add(Relu())
net.add(DenseLayer(20, 1, seed=2000)); net.add(Sigmoid())
net.setupMCMC(0.001, 0.0005, 0.002, 100, 500, 100, 2000, 1, 0.00001, 30, 50, 2, 2)
t0 = time.perf_counter()
rec = net.train(epochs, 10, BernoulliLikelihood(), metricList=[Accuracy()], adjustHypers=True, folderName="MNIST_BNN", networksPerFile=25,
                displaySkip=max(1, epochs // 4))
dt = time.perf_counter() - t0
steps = sum(r["L"] for r in rec)
print(f"kernel {net._chain.kernel_name}: {epochs} epochs, {steps} leapfrog steps in {dt:.1f} s = {dt / epochs * 1e3:.1f} ms per epoch, "
      f"{dt / steps * 1e6:.1f} us per leapfrog step all-in; mean accept {np.mean([r['main']['accept_prob'] for r in rec]):.2f}; "
      f"the tutorial's 2,500 epochs at this rate: {dt / epochs * 2500 / 60:.1f} min")
