"""Likelihood plug-ins (reference: tensorBNN/likelihood.py).

Descriptors for the native library (``kind``, ``fixed_sd``, ``hypers``,
``mainProbsInHypers``) with the reference's class names and keyword
constructors.  ``makeResponseLikelihood`` keeps the reference's keyword
protocol as a NumPy helper for post-processing code; the sampler evaluates the
likelihood inside the fused HIP kernel, never through it.
"""
import math

import numpy as np

from . import _native as nat
from .layer import _multivariate_log_prob


class Likelihood(object):
    kind = None
    fixed_sd = 0.1

    def __init__(self, *argv, **kwargs):
        self.hypers = []
        self.mainProbsInHypers = False

    def makeResponseLikelihood(self, *argv, **kwargs):
        self.hypers = []

    def calcultateLogProb(self, *argv, **kwargs):   # (sic) likelihood.py:98
        pass

    def display(self, hypers):
        pass


class GaussianLikelihood(Likelihood):
    """likelihood.py:63-133: sd is a hyper-parameter stored as sqrt(sd)."""
    kind = nat.LIK_GAUSSIAN

    def __init__(self, *argv, **kwargs):
        self.hypers = [[kwargs["sd"] ** 0.5]]          # :66
        self.mainProbsInHypers = True                  # :67

    def makeResponseLikelihood(self, *argv, **kwargs):
        sd = np.float32(np.asarray(kwargs["hyperStates"][-1]).reshape(-1)[0] ** 2)    # :88
        current = np.asarray(kwargs["predict"](True, argv[0])).T                       # :90-91
        real = np.asarray(kwargs["realVals"], dtype=np.float32).reshape(current.shape)
        return _multivariate_log_prob(np.ones_like(current) * sd, current, real)

    def display(self, hypers):
        print("Loss Standard Deviation: ", float(np.asarray(hypers[-1]).reshape(-1)[0]) ** 2)   # :132


class FixedGaussianLikelihood(Likelihood):
    """likelihood.py:136-202: fixed sd (not squared), no hyper."""
    kind = nat.LIK_FIXED_GAUSSIAN

    def __init__(self, *argv, **kwargs):
        self.hypers = []
        self.sd = kwargs["sd"]
        self.fixed_sd = float(kwargs["sd"])
        self.mainProbsInHypers = False

    def makeResponseLikelihood(self, *argv, **kwargs):
        current = np.asarray(kwargs["predict"](True, argv[0])).T
        real = np.asarray(kwargs["realVals"], dtype=np.float32).reshape(current.shape)
        return _multivariate_log_prob(np.ones_like(current) * np.float32(self.sd), current, real)


class BernoulliLikelihood(Likelihood):
    """likelihood.py:205-243"""
    kind = nat.LIK_BERNOULLI

    def __init__(self, *argv, **kwargs):
        self.hypers = []
        self.mainProbsInHypers = False

    def makeResponseLikelihood(self, *argv, **kwargs):
        p = np.clip(np.asarray(kwargs["predict"](True, argv[0]), dtype=np.float32), 1e-8, 1 - 1e-7)   # :226-231
        y = np.asarray(kwargs["realVals"], dtype=np.float32).reshape(-1, p.shape[0]).T
        return np.where(y == 0, 0, y * np.log(p)) + np.where(1 - y == 0, 0, (1 - y) * np.log1p(-p))

    def calcultateLogProb(self, *argv, **kwargs):
        return [np.float32(0) for _ in range(len(kwargs["hypers"]))]
