"""tensorbnn_amd -- MI355X-native HMC sampler for dense Bayesian neural networks.

Drop-in for the HMC hot path of alpha-davidson/TensorBNN: the same
``network.add() / setupMCMC() / train()`` API and Layer / Likelihood plug-in
surface, with the transition running in hand-written gfx950 HIP kernels behind
the C ABI of ``include/tbnn.h`` (bound in ``_native.py``).  No CPU fallback.
"""
__version__ = "0.1.0"
