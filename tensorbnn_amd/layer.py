"""Layer plug-ins (reference: tensorBNN/layer.py).

Same class names, constructor signatures and attributes as the reference.  For
the HMC path a layer is a *descriptor*: ``network.add`` reads
``inputDims/outputDims/prior_kind/parameters/hypers`` and hands them to the
native library; the sampler never calls the NumPy helper methods below (they
exist for user code written against the reference's plug-in surface, e.g.
``predictor``-style post-processing).
"""
import math

import numpy as np

from . import _native as nat


class Layer(object):
    """Basic layer object (layer.py:10-98)."""

    prior_kind = None      # dense layers: nat.PRIOR_*
    act_kind = None        # activation layers: nat.ACT_*

    def __init__(self, inputDims, outputDims, weights=None, biases=None, activation=None,
                 dtype=np.float32, alpha=0, seed=1):
        self.numTensors = 0
        self.numHyperTensors = 0
        self.inputDims = inputDims
        self.outputDims = outputDims
        self.dtype = dtype
        self.seed = seed
        self.name = "name"

    def calculateProbs(self, *args):
        return np.float32(0.0)

    def calculateHyperProbs(self, hypers, tensors):
        return np.float32(0.0)

    def expand(self, current):
        """rank-2 view of a tensor (layer.py:72-86)."""
        current = np.asarray(current)
        return current.reshape((1,) + current.shape) if current.ndim < 2 else current

    def predict(self, inputTensor, tensors):
        pass


def _mvn1(x, loc, scale):
    """tfd.MultivariateNormalDiag(loc=[loc], scale_diag=[scale]).log_prob([[x]])"""
    z = (x - loc) / scale
    return -0.5 * z * z - math.log(scale) - 0.5 * math.log(2 * math.pi)


def _cauchy_log_prob(gamma, x0, x):
    """BNN_functions.py:37-57 (sign of the first term as in the reference)."""
    x = np.asarray(x, dtype=np.float32)
    return np.log(1 + ((x - x0) / gamma) ** 2) - np.log(np.float32(math.pi * gamma))


def _multivariate_log_prob(sigma, mu, x):
    """BNN_functions.py:7-34 (k = size(sigma))."""
    sigma = np.clip(np.asarray(sigma, dtype=np.float32), 1e-8, 1e8)
    d = (np.asarray(x, dtype=np.float32) - mu) / sigma
    return np.float32(-0.5 * (2 * np.sum(np.log(sigma)) + np.sum(d * d) + sigma.size * math.log(2 * math.pi)))


class _DenseBase(Layer):
    _init_hypers = None
    _name = None

    def __init__(self, inputDims, outputDims, weights=None, biases=None, dtype=np.float32, seed=1):
        self.numTensors = 2          # layer.py:118
        self.numHyperTensors = 4     # layer.py:119
        self.inputDims = inputDims
        self.outputDims = outputDims
        self.dtype = dtype
        self.seed = seed
        self.name = self._name
        self.hypers = np.asarray(self._init_hypers, dtype=np.float32).reshape(4, 1)   # layer.py:156-158
        if weights is None:
            self.parameters = self.sample()
        else:
            self.parameters = [np.asarray(weights, dtype=np.float32).reshape(outputDims, inputDims),
                               np.asarray(biases, dtype=np.float32).reshape(outputDims, 1)]

    def sample(self):
        """N(loc, sqrt(2/out)) initial weights/biases (layer.py:244-264).  TF's
        op-seeded stream is not reproducible; PCG64(seed) / PCG64(seed+1) here."""
        sd = (2.0 / self.outputDims) ** 0.5
        w = np.random.Generator(np.random.PCG64(self.seed)).standard_normal((self.outputDims, self.inputDims))
        b = np.random.Generator(np.random.PCG64(self.seed + 1)).standard_normal((self.outputDims, 1))
        return [(self.hypers[0, 0] + sd * w).astype(np.float32), (self.hypers[2, 0] + sd * b).astype(np.float32)]

    def predict(self, inputTensor, tensors):
        """W @ a + b (layer.py:266-279) -- NumPy helper, not on the HMC path."""
        return self.expand(tensors[0]) @ np.asarray(inputTensor) + self.expand(tensors[1])


class CauchyDenseLayer(_DenseBase):
    """Dense layer with Cauchy priors (layer.py:101-279)."""
    prior_kind = nat.PRIOR_CAUCHY
    _init_hypers = [0.0, 0.5 ** 0.5, 0.0, 0.5 ** 0.5]      # layer.py:132-158
    _name = "dense"

    def calculateProbs(self, hypers, tensors):
        h = np.asarray(hypers, dtype=np.float32).reshape(-1)
        return np.float32(np.sum(_cauchy_log_prob(h[1] ** 2, h[0], tensors[0])) +
                          np.sum(_cauchy_log_prob(h[3] ** 2, h[2], tensors[1])))

    def calculateHyperProbs(self, hypers, tensors):
        h = np.asarray(hypers, dtype=np.float32).reshape(-1)
        p = _mvn1(h[0], 0.0, 0.2) + _mvn1(h[1] ** 2, 0.5 ** 0.5, 0.5) + _mvn1(h[2], 0.0, 0.2) + \
            _mvn1(h[3] ** 2, 0.5 ** 0.5, 0.5)
        return np.float32(p + self.calculateProbs(hypers, tensors))


class GaussianDenseLayer(_DenseBase):
    """Dense layer with Gaussian priors (layer.py:282-459)."""
    prior_kind = nat.PRIOR_GAUSSIAN
    _init_hypers = [0.0, 1.0, 0.0, 1.0]                    # layer.py:316-339
    _name = "denseGaussian"

    def calculateProbs(self, hypers, tensors):
        h = np.asarray(hypers, dtype=np.float32).reshape(-1)
        return np.float32(_multivariate_log_prob(h[1] ** 2, h[0], tensors[0]) +
                          _multivariate_log_prob(h[3] ** 2, h[2], tensors[1]))

    def calculateHyperProbs(self, hypers, tensors):
        h = np.asarray(hypers, dtype=np.float32).reshape(-1)
        p = _mvn1(h[0], 0.0, 0.1) + _mvn1(h[1] ** 2, 1.0, 0.1) + _mvn1(h[2], 0.0, 0.1) + _mvn1(h[3] ** 2, 1.0, 0.1)
        return np.float32(p + self.calculateProbs(hypers, tensors))


DenseLayer = CauchyDenseLayer  # For backwards compatibility (layer.py:461)
