"""Ensemble prediction over saved networks (reference: tensorBNN/predictor.py:15-155).

Reads the reference's on-disk sample format (summary.txt, <n>.<k>.txt,
hypers<k>.txt, architecture.txt; writer network.py:545-663) and runs the
forward pass of every saved network through the native forward kernel.
``trainProbs`` / ``reweight`` (predictor.py:157-273) follow the reference where the
reference runs (likelihood=None: the per-network sum of ``calculateHyperProbs``) and its
intent where it cannot: with a likelihood the reference raises (GaussianLikelihood reads a
``sd`` keyword nobody passes, likelihood.py:116-119; the Fixed-Gaussian one transposes the
rows twice, predictor.py:174 + :141, and never reduces over them) -- here the data term is
the summed log-likelihood of the training rows under each saved network, computed from one
``tbnn_forward_many`` call.  ``autocorrelation`` / ``autoCorrelationLength``
(predictor.py:275-312) use ``emcee.autocorr`` in the reference (un-vendored, emcee 3.x):
``function_1d`` and ``integrated_time`` are restated below from its published algorithm
(FFT autocorrelation; Sokal's automatic window with c = 5).
"""
import math

import numpy as np

from . import _native as nat
from .activationFunctions import Relu, Sigmoid, Tanh
from .layer import CauchyDenseLayer, DenseLayer, GaussianDenseLayer
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
        self.weightsTrain = []                                                  # predictor.py:40
        self.weights = []

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
        self._ensure_chain()
        x = np.asarray(inputMatrix, dtype=np.float32)
        # one native call for the whole ensemble (tbnn_forward_many): the rows are staged once, narrow networks run
        # as one batched launch of the forward-only MFMA kernel
        picked = np.stack([self.vectors[m] for m in range(0, self.numNetworks, n)])
        out = self._chain.forward_many(picked, X=x)
        assert out.shape[0] == math.ceil(self.numNetworks / n)
        return [out[i] for i in range(out.shape[0])]

    # ---- re-weighting (predictor.py:157-273) ----
    def _data_logprob(self, likelihood, trainX, trainY, n):
        """summed log-likelihood of the training rows under every n-th network (see the module docstring)"""
        from .layer import _multivariate_log_prob
        from .likelihood import BernoulliLikelihood, FixedGaussianLikelihood
        preds = self.predict(trainX, n)
        y = np.asarray(trainY, dtype=np.float32)
        out = []
        for i, f in enumerate(preds):
            cur = np.asarray(f, dtype=np.float32).T                             # [rows, d_out]
            real = y.reshape(cur.shape)
            if isinstance(likelihood, BernoulliLikelihood):
                out.append(np.float32(0))                                       # likelihood.py:239-243
                continue
            if isinstance(likelihood, FixedGaussianLikelihood):
                sd = np.float32(likelihood.sd)                                  # likelihood.py:195 (not squared)
            else:
                m = i * n
                sd = np.float32(self.hypers[m][-1]) if len(self.hypers) else np.float32(0.1)   # likelihood.py:116-117
            out.append(np.float32(np.sum(_multivariate_log_prob(np.ones_like(cur) * sd, cur, real))))
        return out

    def _ensure_chain(self):
        if self._chain is None:
            self._chain = nat.Chain(self._descriptor(), likelihood=nat.LIK_FIXED_GAUSSIAN, fixed_sd=1.0,
                                    device=self.device)
            print("tensorbnn_amd: forward kernel", self._chain.kernel_name)    # once; Chain warns when it is the generic one
        return self._chain

    def _hyper_probs(self, weights, n):
        """predictor.py:188-206 / :248-266: minus the data term, minus every layer's calculateHyperProbs.  With the built-in
        dense layers the sum over layers of calculateHyperProbs of EVERY picked network is one launch of the native library
        (tbnn_hyper_probs_many, one workgroup per network, the priors of the architecture loaded right now); layers from
        customLayerDict keep their own Python calculateHyperProbs."""
        picked = list(range(0, self.numNetworks, n))
        dense = [l for l in self.layers if l.numTensors > 0]
        if picked and len(self.hypers) and all(type(l) in (CauchyDenseLayer, GaussianDenseLayer) for l in dense) and \
                all(l.numHyperTensors == 0 for l in self.layers if l.numTensors == 0):
            ch = self._ensure_chain()
            thetas = np.stack([self.vectors[m] for m in picked])
            etas = np.stack([np.asarray(self.hypers[m], dtype=np.float32).reshape(-1)[:4 * len(dense)] for m in picked])
            hp = ch.hyper_probs_many(thetas, etas, priors=[l.prior_kind for l in dense])
            return np.array([np.float32(-np.float32(w) - np.float32(v)) for w, v in zip(weights, hp)])
        for m in picked:
            matrixIndex = hyperIndex = 0
            current = -weights[m // n]
            for layer in self.layers:
                tensors = [self.matrices[matrixIndex + x][m, :, :] for x in range(layer.numTensors)]
                hypers = [self.hypers[m][hyperIndex + x] for x in range(layer.numHyperTensors)]
                hyperIndex += layer.numHyperTensors
                matrixIndex += layer.numTensors
                if layer.numHyperTensors > 0:
                    current = current - np.float32(layer.calculateHyperProbs(hypers, tensors))
            weights[m // n] = current
        return np.array(weights)

    def trainProbs(self, trainX, trainY, n, likelihood):
        """predictor.py:157-207: negative log posterior weight of every n-th network under the training priors"""
        if likelihood is not None:
            weights = self._data_logprob(self.likelihood, trainX, trainY, n)
        else:
            weights = [np.float32(0) for _ in range(0, self.numNetworks, n)]
        self.weightsTrain = self._hyper_probs(weights, n)

    def reweight(self, architecture, trainX=None, trainY=None, n=1, likelihood=None):
        """predictor.py:209-273: importance weights p(theta | new priors) / p(theta | training priors), normalised"""
        if len(self.weightsTrain) == 0:
            self.trainProbs(trainX, trainY, n, likelihood)
        self.loadArchitecture(architecture=architecture)
        if likelihood is not None:
            weights = self._data_logprob(likelihood, trainX, trainY, n)
        else:
            weights = [np.float32(0) for _ in range(0, self.numNetworks, n)]
        self.weights = self._hyper_probs(weights, n)
        weighting = np.exp(self.weightsTrain - self.weights)
        weighting = weighting / np.sum(weighting)
        self.loadArchitecture()
        return weighting

    # ---- autocorrelation (predictor.py:275-312) ----
    def autocorrelation(self, inputData, nMax):
        output = np.squeeze(np.array(self.predict(inputData, n=1))).T
        valFunc, accepted = 0, 0
        for x in range(len(output)):
            temp = integrated_time(output[x], tol=5, quiet=True)
            if not np.isnan(temp).any():
                valFunc = valFunc + np.array(function_1d(output[x]))
                accepted += 1
        valFunc = valFunc / accepted
        return valFunc[:nMax] if nMax < len(valFunc) else valFunc

    def autoCorrelationLength(self, inputData, nMax):
        output = np.squeeze(np.array(self.predict(inputData, n=1))).T
        val, accepted = 0, 0
        for x in range(len(output)):
            temp = integrated_time(output[x], tol=5, quiet=True)
            if not np.isnan(temp).any():
                val = val + temp
                accepted += 1
        val = val / accepted
        if val[0] > nMax:
            print("Correlation time is greater than maximum accepted value.")
        return val[0]

    def extractParameters(self):
        """predictor.py:314-319: the parameter matrices, first axis = network"""
        return self.matrices


def function_1d(x):
    """emcee.autocorr.function_1d: normalised autocorrelation function of a 1-D series (FFT, zero-padded)"""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if x.ndim != 1:
        raise ValueError("invalid dimensions for 1D autocorrelation function")
    n = 1
    while n < len(x):
        n <<= 1
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def integrated_time(x, c=5, tol=50, quiet=False):
    """emcee.autocorr.integrated_time for one chain: tau = 2 * cumsum(acf) - 1 at Sokal's window (first M >= c * tau);
    returns an array of one element, NaN-free unless the series is constant"""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    f = function_1d(x)
    taus = 2.0 * np.cumsum(f) - 1.0
    m = np.arange(len(taus)) < c * taus
    window = int(np.argmin(m)) if np.any(m) else len(taus) - 1
    tau_est = np.array([taus[window]])
    if not quiet and np.any(tol * tau_est > len(x)):
        raise ValueError("The chain is shorter than {0} times the integrated autocorrelation time".format(tol))
    return tau_est
