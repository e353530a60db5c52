"""Activation plug-ins (reference: tensorBNN/activationFunctions.py).

Relu / Sigmoid / Tanh / Exp / Elu are fused into the native kernels as the epilogue
of the dense layer they follow.  The remaining reference activations (Softmax,
Leaky_relu, Prelu, SquarePrelu) are outside this build's hot-path
scope: ``network.add`` rejects them loudly.
"""
import numpy as np

from . import _native as nat
from .layer import Layer


class _Activation(Layer):
    act_kind = None
    _name = None

    def __init__(self, inputDims=None, outputDims=None):
        self.numTensors = 0
        self.numHyperTensors = 0
        self.name = self._name


class Relu(_Activation):
    """activationFunctions.py:27-37"""
    act_kind = nat.ACT_RELU
    _name = "relu"

    def predict(self, inputTensor, _):
        return np.maximum(np.asarray(inputTensor), 0)


class Sigmoid(_Activation):
    """activationFunctions.py:40-50"""
    act_kind = nat.ACT_SIGMOID
    _name = "sigmoid"

    def predict(self, inputTensor, _):
        return 1.0 / (1.0 + np.exp(-np.asarray(inputTensor)))


class Tanh(_Activation):
    """activationFunctions.py:53-63"""
    act_kind = nat.ACT_TANH
    _name = "tanh"

    def predict(self, inputTensor, _):
        return np.tanh(np.asarray(inputTensor))


class _Unsupported(_Activation):
    def __init__(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__}: not covered by the MI355X HMC path (SURVEY.md section 8(f) rank 3); "
            "supported activations: Relu, Sigmoid, Tanh, Exp, Elu")


class Exp(_Activation):
    """activationFunctions.py:14-24"""
    act_kind = nat.ACT_EXP
    _name = "Exp"

    def predict(self, inputTensor, _):
        return np.exp(np.asarray(inputTensor))


class Elu(_Activation):
    """activationFunctions.py:66-76"""
    act_kind = nat.ACT_ELU
    _name = "elu"

    def predict(self, inputTensor, _):
        z = np.asarray(inputTensor)
        return np.where(z > 0, z, np.expm1(z))


class Softmax(_Unsupported):
    _name = "softmax"


class Leaky_relu(_Unsupported):
    _name = "leakyrelu"


class Prelu(_Unsupported):
    _name = "prelu"


class SquarePrelu(_Unsupported):
    _name = "squareprelu"
