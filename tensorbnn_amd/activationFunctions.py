"""Activation plug-ins (reference: tensorBNN/activationFunctions.py).

Relu / Sigmoid / Tanh / Exp / Elu are fused into the native kernels as the epilogue
of the dense layer they follow.  The remaining reference activations are rejected
loudly by ``network.add`` -- none of them has a reference behaviour on the HMC path
that a parity test could pin:

* ``Prelu`` / ``SquarePrelu`` (activationFunctions.py:117-433) cannot be trained in the
  reference: the target closure calls ``layer.calculateProbs(hypers, tensors)``
  (network.py:300-303, 380-383) but both define ``calculateProbs(self, slopes)``
  (activationFunctions.py:178, 330) -- a TypeError on the first HMC step.
* ``Leaky_relu`` (activationFunctions.py:92-114) puts its fixed ``alpha`` into the sampled
  state with no prior and, having no hyper tensors, never advances the tensor index
  (quirk Q7, network.py:300-306): every later layer's prior is evaluated on shifted
  tensors.
* ``Softmax`` (activationFunctions.py:79-89) normalises over the *last* axis of the
  ``[units, rows]`` tensor, i.e. across the data rows, not across the units.
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
    _why = "see activationFunctions.py's module docstring"

    def __init__(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__}: not covered by the MI355X HMC path -- {self._why}; "
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
    _why = "the reference's softmax runs over the data rows (last axis of [units, rows], activationFunctions.py:88)"


class Leaky_relu(_Unsupported):
    _name = "leakyrelu"
    _why = "the reference samples alpha as a prior-less state and mis-indexes every later layer's prior (network.py:300-306)"


class Prelu(_Unsupported):
    _name = "prelu"
    _why = "the reference's Prelu.calculateProbs(self, slopes) does not accept the (hypers, tensors) call of network.py:300-303"


class SquarePrelu(_Unsupported):
    _name = "squareprelu"
    _why = "the reference's SquarePrelu.calculateProbs(self, slopes) does not accept the (hypers, tensors) call of network.py:300-303"
