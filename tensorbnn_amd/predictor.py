"""Ensemble prediction over saved networks (reference: tensorBNN/predictor.py:15-155).

Reads the reference's on-disk sample format (summary.txt, <n>.<k>.txt,
hypers<k>.txt, architecture.txt; writer network.py:545-663) and runs the
forward pass of every saved network through the native forward kernel.
Re-weighting / autocorrelation (predictor.py:157-312, needs ``emcee``) are out
of this build's scope.
"""
import math

import numpy as np

from . import _native as nat
from .activationFunctions import Relu, Sigmoid, Tanh
from .layer import DenseLayer, GaussianDenseLayer
from .likelihood import GaussianLikelihood


class predictor(object):
    def __init__(self, directoryPath, dtype=np.float32, customLayerDict={}, likelihood=None, device=0):
        self.layerDict = {"relu": Relu, "sigmoid": Sigmoid, "tanh": Tanh, "dense": DenseLayer,
                          "denseGaussian": GaussianDenseLayer}                 # predictor.py:30-36
        self.layerDict.update(customLayerDict)
        self.directoryPath = directoryPath if directoryPath.endswith("/") else directoryPath + "/"
        self.dtype = np.float32
        self.device = device
        self.loadNetworks()
        self.loadArchitecture()
        self.likelihood = likelihood if likelihood is not None else GaussianLikelihood(sd=0.1)
        self._chain = None

    def loadNetworks(self):
        """predictor.py:43-113"""
        summary = []
        with open(self.directoryPath + "summary.txt", "r") as file:
            for line in iter(file):
                summary.append(line.split())
        numNetworks = int(summary[-2][0])
        numFiles = int(summary[-2][1])
        numMatrices = int(summary[-2][2])
        numHypers = int(summary[-1][0])
        numNetworks //= numFiles
        matrices = []
        for n in range(numMatrices):
            d1 = int(summary[n][0])
            d2 = int(summary[n][1]) if len(summary[n]) == 2 else 1
            weights0 = np.zeros((numNetworks * numFiles, d1, d2), dtype=np.float32)
            for m in range(numFiles):
                weights = np.loadtxt(self.directoryPath + str(n) + "." + str(m) + ".txt", dtype=np.float32, ndmin=2)
                for k in range(numNetworks):
                    weights0[m * numNetworks + k, :, :] = weights[d1 * k:d1 * (k + 1), :d2]
            matrices.append(weights0)
        hypers = []
        if numHypers > 0:
            for m in range(numFiles):
                weights = np.loadtxt(self.directoryPath + "hypers" + str(m) + ".txt", dtype=np.float32, ndmin=1)
                for k in range(numNetworks):
                    hypers.append(weights[numHypers * k:numHypers * (k + 1)])
        self.numNetworks = numNetworks * numFiles
        self.numMatrices = numMatrices
        self.matrices = matrices
        self.hypers = hypers
        self.vectors = [np.concatenate([mat[i].reshape(-1) for mat in matrices]) for i in range(self.numNetworks)]

    def loadArchitecture(self, architecture=None):
        """predictor.py:115-130: layer classes looked up by name, built with (1, 1)."""
        path = self.directoryPath + "architecture.txt" if architecture is None else architecture
        self.layers = []
        with open(path, "r") as file:
            for line in iter(file):
                self.layers.append(self.layerDict[line.replace("\n", "")](inputDims=1, outputDims=1))

    def _descriptor(self):
        dense, mi = [], 0
        for layer in self.layers:
            if layer.numTensors > 0:
                out_dim, in_dim = self.matrices[mi].shape[1], self.matrices[mi].shape[2]
                dense.append([in_dim, out_dim, nat.ACT_NONE, layer.prior_kind])
                mi += layer.numTensors
            else:
                dense[-1][2] = layer.act_kind
        return dense

    def predict(self, inputMatrix, n=1):
        """predictor.py:132-155: list of [d_out, rows] predictions, every n-th network."""
        if self._chain is None:
            self._chain = nat.Chain(self._descriptor(), likelihood=nat.LIK_FIXED_GAUSSIAN, fixed_sd=1.0,
                                    device=self.device)
        x = np.asarray(inputMatrix, dtype=np.float32)
        # one native call for the whole ensemble (tbnn_forward_many): the rows are staged once, narrow networks run
        # as one batched launch of the forward-only MFMA kernel
        picked = np.stack([self.vectors[m] for m in range(0, self.numNetworks, n)])
        out = self._chain.forward_many(picked, X=x)
        assert out.shape[0] == math.ceil(self.numNetworks / n)
        return [out[i] for i in range(out.shape[0])]
