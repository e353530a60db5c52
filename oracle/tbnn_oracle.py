"""CPU oracle for the TensorBNN HMC hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-NumPy restatement of the algorithm the reference
(alpha-davidson/TensorBNN, /root/reference, Python + TensorFlow-Probability)
runs on its HMC hot path.  It exists so that the HIP kernels in
``tensorbnn_amd/csrc`` can be checked against *something that follows the
reference line by line*.  It is never imported by the product package: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may use it.

PARITY UNPINNED.  The reference cannot be imported here (TensorFlow and
TensorFlow-Probability are not installed, there is no network) and it ships no
tests, golden vectors or fixtures.  Part of the path (leapfrog, Metropolis,
autodiff) lives in the un-vendored third-party package
``tensorflow-probability`` (0.12.2 per reference README.md:32; 0.11 per
docs/Setup.md:21) on ``tensorflow`` 2.5 (README.md:26); its published
algorithm (``tfp.mcmc.HamiltonianMonteCarlo`` with the
``SimpleLeapfrogIntegrator``) is restated in :func:`hmc_step`.  What pins this
oracle instead (see tests/test_oracle.py):
  * closed-form known-answer tests against scipy.stats for the two density
    helpers, with the reference's quirks (Q1 sign, Q2 normaliser) encoded and
    the delta to scipy asserted;
  * torch.autograd (fp64) gradients of the restated log-density for every
    hand-coded reverse-mode formula, for weights and for hyper-parameters;
  * HMC invariants (reversibility, energy error O(eps^2), eps->0 => accept->1).

Every function cites the reference file:line it restates.  ``dtype`` selects
the arithmetic: ``np.float32`` mimics the reference op by op (it computes in
tf.float32 everywhere: Examples/trainRegression.py:41, layer.py:116,
paramAdapter.py:60); ``np.float64`` is the high-precision arm used to bound
rounding error.
"""
from __future__ import annotations

import math
import os
import random as _pyrandom
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# Descriptors (the oracle's own; deliberately independent of the product's)
# ----------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_EXP, ACT_ELU = 0, 1, 2, 3, 4, 5
PRIOR_CAUCHY, PRIOR_GAUSSIAN = 0, 1
LIK_GAUSSIAN, LIK_FIXED_GAUSSIAN, LIK_BERNOULLI = 0, 1, 2


@dataclass
class DenseSpec:
    """One dense layer + the activation layer that follows it (if any).

    reference: layer.py:101-279 (CauchyDenseLayer), :282-459
    (GaussianDenseLayer); activationFunctions.py:27-63.
    """
    in_dim: int
    out_dim: int
    act: int = ACT_NONE
    prior: int = PRIOR_CAUCHY


@dataclass
class NetSpec:
    layers: List[DenseSpec]
    likelihood: int = LIK_GAUSSIAN
    fixed_sd: float = 0.1          # FixedGaussianLikelihood(sd=...) likelihood.py:138-141

    @property
    def n_params(self) -> int:
        return sum(l.in_dim * l.out_dim + l.out_dim for l in self.layers)

    @property
    def n_hypers(self) -> int:
        # 4 per dense layer (network.py:189-191, layer.py:156-158) + sqrt(sd)
        # for GaussianLikelihood (likelihood.py:66, network.py:542-543)
        return 4 * len(self.layers) + (1 if self.likelihood == LIK_GAUSSIAN else 0)

    def offsets(self) -> List[Tuple[int, int]]:
        """(offset of W, offset of b) of each dense layer inside theta."""
        out, o = [], 0
        for l in self.layers:
            out.append((o, o + l.in_dim * l.out_dim))
            o += l.in_dim * l.out_dim + l.out_dim
        return out


def default_hypers(spec: NetSpec, sd: float = 0.1, dtype=np.float32) -> np.ndarray:
    """Initial eta.  Cauchy: [x0_w=0, g_w=sqrt(.5), x0_b=0, g_b=sqrt(.5)]
    (layer.py:136-158); Gaussian: [0, 1, 0, 1] (layer.py:316-339); then
    sqrt(sd) for GaussianLikelihood (likelihood.py:66)."""
    eta = []
    for l in spec.layers:
        if l.prior == PRIOR_CAUCHY:
            eta += [0.0, 0.5 ** 0.5, 0.0, 0.5 ** 0.5]
        else:
            eta += [0.0, 1.0, 0.0, 1.0]
    if spec.likelihood == LIK_GAUSSIAN:
        eta.append(sd ** 0.5)
    return np.asarray(eta, dtype=dtype)


def unflatten(spec: NetSpec, theta: np.ndarray):
    """theta -> [(W[out,in], b[out,1]), ...] (network.add order, network.py:173-191)."""
    parts = []
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        W = theta[ow:ow + l.in_dim * l.out_dim].reshape(l.out_dim, l.in_dim)
        b = theta[ob:ob + l.out_dim].reshape(l.out_dim, 1)
        parts.append((W, b))
    return parts


def flatten(parts) -> np.ndarray:
    return np.concatenate([np.concatenate([W.reshape(-1), b.reshape(-1)]) for W, b in parts])


# ----------------------------------------------------------------------------
# L0 math helpers  (reference: tensorBNN/BNN_functions.py)
# ----------------------------------------------------------------------------
def multivariate_log_prob(sigma_in, mu, x, dtype=np.float32):
    """BNN_functions.py:7-34.  Note k = size(sigma) (Q2): with a scalar sigma
    and an N-element x the normaliser is counted once, not N times."""
    dt = dtype
    sigma = np.asarray(sigma_in, dtype=dt)
    sigma = np.maximum(sigma, dt(10 ** (-8)))                      # :23
    sigma = np.minimum(sigma, dt(10 ** 8))                         # :24
    log_det = dt(2) * np.sum(np.log(sigma), dtype=dt)              # :25
    k = dt(sigma.size)                                             # :26
    inv = dt(1) / sigma                                            # :27
    dif_sigma = inv * (np.asarray(x, dtype=dt) - np.asarray(mu, dtype=dt))   # :28
    dif_sigma_sq = np.sum(dif_sigma * dif_sigma, dtype=dt)         # :29
    two_pi = dt(2 * math.pi)                                       # :30
    return dt(-0.5) * (log_det + dif_sigma_sq + k * np.log(two_pi))  # :32


def cauchy_log_prob(gamma, x0, x, dtype=np.float32):
    """BNN_functions.py:37-57.  Returns +log(1+z^2) - log(pi*gamma) per
    element (Q1: the sign of the first term is the reference's, not Cauchy's)."""
    dt = dtype
    gamma = dt(gamma)
    x0 = dt(x0)
    x = np.asarray(x, dtype=dt)
    a = np.log(dt(1) + ((x - x0) / gamma) ** 2)                    # :51
    b = np.log(dt(dt(math.pi) * gamma))                            # :52
    return (a + (-b) * np.ones_like(x)).astype(dt)                 # :53-56


def mvn_diag_scalar_log_prob(x, loc, scale, dtype=np.float32):
    """tfd.MultivariateNormalDiag(loc=[loc], scale_diag=[scale]).log_prob([[x]])
    (layer.py:137-153, :221-228): -0.5*((x-loc)/scale)^2 - log(scale) - 0.5*log(2pi)."""
    dt = dtype
    z = (dt(x) - dt(loc)) / dt(scale)
    return dt(-0.5) * z * z - np.log(dt(scale)) - dt(0.5) * np.log(dt(2 * math.pi))


# ----------------------------------------------------------------------------
# L3 plug-ins: activations, dense forward, priors, likelihoods
# ----------------------------------------------------------------------------
def activate(z, act):
    """activationFunctions.py:35-37 (relu), :48-50 (sigmoid), :61-63 (tanh)."""
    if act == ACT_NONE:
        return z
    if act == ACT_RELU:
        return np.maximum(z, z.dtype.type(0))
    if act == ACT_TANH:
        return np.tanh(z)
    if act == ACT_SIGMOID:
        return (z.dtype.type(1) / (z.dtype.type(1) + np.exp(-z))).astype(z.dtype)
    if act == ACT_EXP:                      # activationFunctions.py:23
        return np.exp(z)
    if act == ACT_ELU:                      # activationFunctions.py:75 (gen_nn_ops.elu)
        return np.where(z > 0, z, np.expm1(z)).astype(z.dtype)
    raise ValueError(act)


def act_grad_from_output(a, act):
    """d act / d z expressed through the activation's output a."""
    one = a.dtype.type(1)
    if act == ACT_NONE:
        return np.ones_like(a)
    if act == ACT_RELU:
        return (a > 0).astype(a.dtype)
    if act == ACT_TANH:
        return one - a * a
    if act == ACT_SIGMOID:
        return a * (one - a)
    if act == ACT_EXP:
        return a
    if act == ACT_ELU:
        return np.where(a > 0, one, a + one).astype(a.dtype)
    raise ValueError(act)


def forward(spec: NetSpec, theta, X, dtype=np.float32, keep=False):
    """network.predict / innerPrediction, network.py:141-171; dense
    layer.py:266-279: W[out,in] @ a[in,n] + b[out,1].  Returns f[d_out, n]
    (and the list of layer outputs a_0..a_L when keep=True)."""
    a = np.ascontiguousarray(np.asarray(X, dtype=dtype).T)         # :161 transpose
    acts = [a]
    for l, (W, b) in zip(spec.layers, unflatten(spec, np.asarray(theta, dtype=dtype))):
        z = W @ a + b                                              # layer.py:278
        a = activate(z, l.act)
        acts.append(a)
    return (a, acts) if keep else a


def layer_log_prob(l: DenseSpec, h4, W, b, dtype=np.float32):
    """calculateProbs: layer.py:166-197 (Cauchy), :346-377 (Gaussian)."""
    dt = dtype
    h4 = np.asarray(h4, dtype=dt)
    loc_w, scale_w, loc_b, scale_b = h4[0], h4[1] ** 2, h4[2], h4[3] ** 2   # :177-180 / :357-360 (Q3)
    if l.prior == PRIOR_CAUCHY:
        p = np.sum(cauchy_log_prob(scale_w, loc_w, W, dt), dtype=dt)
        p = p + np.sum(cauchy_log_prob(scale_b, loc_b, b, dt), dtype=dt)
        return dt(p)
    p = multivariate_log_prob(scale_w, loc_w, W, dt)
    p = p + multivariate_log_prob(scale_b, loc_b, b, dt)
    return dt(p)


def layer_hyper_log_prob(l: DenseSpec, h4, W, b, dtype=np.float32):
    """calculateHyperProbs: layer.py:199-242 (Cauchy), :379-422 (Gaussian).
    Hyper-priors are evaluated at the *squared* value (Q4)."""
    dt = dtype
    h4 = np.asarray(h4, dtype=dt)
    loc_w, scale_w, loc_b, scale_b = h4[0], h4[1] ** 2, h4[2], h4[3] ** 2
    if l.prior == PRIOR_CAUCHY:
        p = mvn_diag_scalar_log_prob(loc_w, 0.0, 0.2, dt)            # :136-138, :221
        p = p + mvn_diag_scalar_log_prob(scale_w, 0.5 ** 0.5, 0.5, dt)   # :141-143, :223
        p = p + mvn_diag_scalar_log_prob(loc_b, 0.0, 0.2, dt)        # :146-148, :226
        p = p + mvn_diag_scalar_log_prob(scale_b, 0.5 ** 0.5, 0.5, dt)   # :151-153, :228
    else:
        p = mvn_diag_scalar_log_prob(loc_w, 0.0, 0.1, dt)            # :317-319, :401
        p = p + mvn_diag_scalar_log_prob(scale_w, 1.0, 0.1, dt)      # :322-324, :403
        p = p + mvn_diag_scalar_log_prob(loc_b, 0.0, 0.1, dt)        # :327-329, :406
        p = p + mvn_diag_scalar_log_prob(scale_b, 1.0, 0.1, dt)      # :332-334, :408
    return dt(p + layer_log_prob(l, h4, W, b, dt))


def likelihood_sigma(spec: NetSpec, eta, dtype=np.float32):
    """sd used by the Gaussian likelihoods: eta[-1]**2 (likelihood.py:88, Q3)
    or the fixed sd un-squared (likelihood.py:162)."""
    if spec.likelihood == LIK_GAUSSIAN:
        return dtype(np.asarray(eta, dtype=dtype)[-1] ** 2)
    return dtype(spec.fixed_sd)


def log_likelihood(spec: NetSpec, eta, f, Y, dtype=np.float32):
    """makeResponseLikelihood summed (network.py:388-391): Gaussian
    likelihood.py:69-96, fixed :143-169, Bernoulli :210-237.  f is [d_out,n]."""
    dt = dtype
    f = np.asarray(f, dtype=dt)
    if spec.likelihood in (LIK_GAUSSIAN, LIK_FIXED_GAUSSIAN):
        cur = f.T                                                   # :91
        sigma = np.ones_like(cur) * likelihood_sigma(spec, eta, dt)  # :92
        real = np.asarray(Y, dtype=dt).reshape(cur.shape)           # :93
        return multivariate_log_prob(sigma, cur, real, dt)          # :94
    # Bernoulli: clip :226-231; tfd.Bernoulli(probs).log_prob(y) =
    # xlogy(y,p) + xlog1py(1-y,-p) (TFP 0.12 bernoulli.py _log_prob)
    p = np.clip(f, dt(1e-8), dt(1 - 1e-7)).astype(dt)
    y = np.asarray(Y, dtype=dt).reshape(-1, f.shape[0]).T           # :236 transpose(realVals)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = np.where(y == 0, dt(0), y * np.log(p))
        t2 = np.where((dt(1) - y) == 0, dt(0), (dt(1) - y) * np.log1p(-p))
    return dt(np.sum((t1 + t2).astype(dt), dtype=dt))


def target_log_prob(spec: NetSpec, theta, eta, X, Y, dtype=np.float32):
    """The closure calculateProbs of network.py:370-392 (= :290-314)."""
    dt = dtype
    theta = np.asarray(theta, dtype=dt)
    eta = np.asarray(eta, dtype=dt)
    prob = dt(0)
    for i, (l, (W, b)) in enumerate(zip(spec.layers, unflatten(spec, theta))):
        prob = dt(prob + layer_log_prob(l, eta[4 * i:4 * i + 4], W, b, dt))   # :382-384
    f = forward(spec, theta, X, dt)
    return dt(prob + log_likelihood(spec, eta, f, Y, dt))            # :388-391


# ----------------------------------------------------------------------------
# A12: hand-coded reverse mode (TF autodiff restated; checked vs torch.autograd)
# ----------------------------------------------------------------------------
def prior_grad(l: DenseSpec, h4, W, b, dtype=np.float32):
    dt = dtype
    h4 = np.asarray(h4, dtype=dt)
    out = []
    for x, loc, g in ((W, h4[0], h4[1]), (b, h4[2], h4[3])):
        scale = g ** 2
        if l.prior == PRIOR_CAUCHY:
            z = (x - loc) / scale
            out.append((dt(2) * z / (scale * (dt(1) + z * z))).astype(dt))   # d/dx +log(1+z^2)
        else:
            s = min(max(scale, dt(1e-8)), dt(1e8))
            out.append((-(x - loc) / (s * s)).astype(dt))
    return out


def target_log_prob_and_grad(spec: NetSpec, theta, eta, X, Y, dtype=np.float32):
    """value_and_gradient of the target w.r.t. theta (what TFP asks TF autodiff
    for inside the leapfrog loop; call sites network.py:394-408)."""
    dt = dtype
    theta = np.asarray(theta, dtype=dt)
    eta = np.asarray(eta, dtype=dt)
    parts = unflatten(spec, theta)
    f, acts = forward(spec, theta, X, dt, keep=True)
    logp = target_log_prob(spec, theta, eta, X, Y, dt)
    n = f.shape[1]
    # dL/df  [d_out, n]
    if spec.likelihood in (LIK_GAUSSIAN, LIK_FIXED_GAUSSIAN):
        sig = likelihood_sigma(spec, eta, dt)
        sig = min(max(sig, dt(1e-8)), dt(1e8))
        y = np.asarray(Y, dtype=dt).reshape(n, -1).T
        d_a = ((y - f) / (sig * sig)).astype(dt)
    else:
        y = np.asarray(Y, dtype=dt).reshape(-1, f.shape[0]).T
        inside = (f > dt(1e-8)) & (f < dt(1 - 1e-7))
        p = np.clip(f, dt(1e-8), dt(1 - 1e-7)).astype(dt)
        d_a = np.where(inside, y / p - (dt(1) - y) / (dt(1) - p), dt(0)).astype(dt)
    grads = [None] * len(spec.layers)
    for i in range(len(spec.layers) - 1, -1, -1):
        l = spec.layers[i]
        W, b = parts[i]
        delta = (d_a * act_grad_from_output(acts[i + 1], l.act)).astype(dt)   # dL/dz_i
        gW = delta @ acts[i].T
        gb = np.sum(delta, axis=1, keepdims=True, dtype=dt)
        pW, pb = prior_grad(l, eta[4 * i:4 * i + 4], W, b, dt)
        grads[i] = ((gW + pW).astype(dt), (gb + pb).astype(dt))
        if i > 0:
            d_a = (W.T @ delta).astype(dt)
    return logp, flatten(grads).astype(dt)


# ----------------------------------------------------------------------------
# A11/A13: tfp.mcmc.HamiltonianMonteCarlo + sample_chain(num_results=1)
# ----------------------------------------------------------------------------
@dataclass
class StepResult:
    theta: np.ndarray
    accepted: bool
    log_accept_ratio: float
    accept_prob: float
    logp_old: float
    logp_new: float
    sjd: float
    trace_logp: List[float] = field(default_factory=list)
    theta_proposed: Optional[np.ndarray] = None
    p_final: Optional[np.ndarray] = None


def hmc_step(value_and_grad, q0, eps, L, p0, log_u, dtype=np.float32, energy_dtype=None):
    """One HMC transition, TFP 0.12 ordering (un-vendored; restated from the
    published algorithm; call sites network.py:315-329, :394-408):

      bootstrap_results: (logp0, g0) = value_and_grad(q0)
      p = p0 + (eps/2) g0                               (half kick)
      L x { q = q + eps p ; (logp, g) = value_and_grad(q) ; p = p + eps g }
      p = p - (eps/2) g                                 (undo half kick)
      log_accept_ratio = logp_L - logp_0 + 1/2|p0|^2 - 1/2|p_L|^2  (non-finite -> -inf)
      accept iff log_u < log_accept_ratio

    ``energy_dtype`` (default = dtype) is the type the scalar energies are
    summed/differenced in; the HIP path sums in float64.
    acceptRate = lar<0 ? exp(lar) : 1   (network.py:410-411).
    """
    dt = dtype
    et = energy_dtype or dtype
    q0 = np.asarray(q0, dtype=dt)
    p0 = np.asarray(p0, dtype=dt)
    eps = dt(eps)
    logp0, g = value_and_grad(q0)
    trace = [float(logp0)]
    p = (p0 + dt(0.5) * eps * g).astype(dt)
    q = q0.copy()
    logp = logp0
    for _ in range(int(L)):
        q = (q + eps * p).astype(dt)
        logp, g = value_and_grad(q)
        trace.append(float(logp))
        p = (p + eps * g).astype(dt)
    p = (p - dt(0.5) * eps * g).astype(dt)
    k0 = et(0.5) * np.sum(p0.astype(et) ** 2, dtype=et)
    k1 = et(0.5) * np.sum(p.astype(et) ** 2, dtype=et)
    lar = et(logp) - et(logp0) + k0 - k1
    if not np.isfinite(lar):
        lar = et(-np.inf)
    accepted = bool(et(log_u) < lar)
    acc_prob = float(np.exp(lar)) if lar < 0 else 1.0
    new = q if accepted else q0
    sjd = float(np.sum((new.astype(np.float64) - q0.astype(np.float64)) ** 2))
    return StepResult(theta=new.copy(), accepted=accepted, log_accept_ratio=float(lar),
                      accept_prob=acc_prob, logp_old=float(logp0), logp_new=float(logp),
                      sjd=sjd, trace_logp=trace, theta_proposed=q.copy(), p_final=p.copy())


def weight_step(spec, theta, eta, X, Y, eps, L, p0, log_u, dtype=np.float32, energy_dtype=None):
    """InnerStepMain, network.py:368-412."""
    vg = lambda q: target_log_prob_and_grad(spec, q, eta, X, Y, dtype)
    return hmc_step(vg, theta, eps, L, p0, log_u, dtype, energy_dtype)


# ----------------------------------------------------------------------------
# A14: hyper-parameter transition + dual averaging (network.py:414-471)
# ----------------------------------------------------------------------------
def hyper_log_prob(spec: NetSpec, eta, theta, X, Y, dtype=np.float32):
    """closure calculateProbs of InnerStepHyper, network.py:416-440."""
    dt = dtype
    eta = np.asarray(eta, dtype=dt)
    theta = np.asarray(theta, dtype=dt)
    prob = dt(0)
    for i, (l, (W, b)) in enumerate(zip(spec.layers, unflatten(spec, theta))):
        prob = dt(prob + layer_hyper_log_prob(l, eta[4 * i:4 * i + 4], W, b, dt))   # :429-431
    if spec.likelihood == LIK_GAUSSIAN:                              # mainProbsInHypers, likelihood.py:67
        f = forward(spec, theta, X, dt)
        prob = dt(prob + log_likelihood(spec, eta, f, Y, dt))        # :435-438
    return prob


def hyper_log_prob_and_grad(spec: NetSpec, eta, theta, X, Y, dtype=np.float32, S=None):
    """Hand-coded gradient of :func:`hyper_log_prob` w.r.t. eta (checked vs
    torch.autograd in tests).  ``S`` = sum((y-f)^2) may be supplied (SURVEY
    section 7.3 closed form); the value then uses
    -1/2 (2 n log s + S/s^2 + n log 2pi)."""
    dt = dtype
    eta = np.asarray(eta, dtype=dt)
    theta = np.asarray(theta, dtype=dt)
    g = np.zeros_like(eta)
    parts = unflatten(spec, theta)
    prob = dt(0)
    for i, (l, (W, b)) in enumerate(zip(spec.layers, parts)):
        h4 = eta[4 * i:4 * i + 4]
        prob = dt(prob + layer_hyper_log_prob(l, h4, W, b, dt))
        for j, x in ((0, W), (2, b)):
            loc, gg = h4[j], h4[j + 1]
            scale = gg * gg
            if l.prior == PRIOR_CAUCHY:
                z = (x - loc) / scale
                w = dt(2) * z / (dt(1) + z * z)
                d_loc = np.sum(-w / scale, dtype=dt)
                d_scale = np.sum(-w * z / scale - dt(1) / scale, dtype=dt)
                d_loc += -(loc - dt(0)) / dt(0.2) ** 2
                d_scale += -(scale - dt(0.5 ** 0.5)) / dt(0.5) ** 2
            else:
                s = min(max(scale, dt(1e-8)), dt(1e8))
                clamped = not (dt(1e-8) < scale < dt(1e8))
                d_loc = np.sum((x - loc) / (s * s), dtype=dt)
                d_scale = dt(0) if clamped else (-dt(1) / s + np.sum((x - loc) ** 2, dtype=dt) / (s * s * s))
                d_loc += -(loc - dt(0)) / dt(0.1) ** 2
                d_scale += -(scale - dt(1.0)) / dt(0.1) ** 2
            g[4 * i + j] = d_loc
            g[4 * i + j + 1] = d_scale * dt(2) * gg
    if spec.likelihood == LIK_GAUSSIAN:
        s_raw = eta[-1] ** 2
        s = min(max(s_raw, dt(1e-8)), dt(1e8))
        clamped = not (dt(1e-8) < s_raw < dt(1e8))
        f = None
        if S is None:
            f = forward(spec, theta, X, dt)
            n_el = f.size
            y = np.asarray(Y, dtype=dt).reshape(f.shape[1], -1).T
            S_ = np.sum((y - f).astype(np.float64) ** 2)
            prob = dt(prob + log_likelihood(spec, eta, f, Y, dt))
        else:
            S_ = float(S)
            n_el = int(np.asarray(Y).size)
            prob = dt(prob + dt(-0.5) * (dt(2 * n_el) * np.log(s) + dt(S_) / (s * s)
                                          + dt(n_el) * np.log(dt(2 * math.pi))))
        d_s = dt(0) if clamped else dt(-n_el / float(s) + S_ / float(s) ** 3)
        g[-1] = d_s * dt(2) * eta[-1]
    return prob, g


def hyper_step(spec, eta, theta, X, Y, eps_h, L_h, p0, log_u, dtype=np.float32, S=None,
               energy_dtype=None):
    """InnerStepHyper's HMC part, network.py:442-456."""
    vg = lambda e: hyper_log_prob_and_grad(spec, e, theta, X, Y, dtype, S=S)
    return hmc_step(vg, eta, eps_h, L_h, p0, log_u, dtype, energy_dtype)


@dataclass
class DualAveragingState:
    """setupMCMC constants, network.py:241-248."""
    hyper_step_size: float
    burnin: int
    target: float = 0.95
    gamma: float = 0.4
    t0: float = 10.0
    kappa: float = 0.75
    h: float = 0.0
    log_eps_bar: float = 0.0
    mu: float = 0.0

    def __post_init__(self):
        self.mu = float(np.log(np.float32(100 * self.hyper_step_size)))      # :248 (Q6)
        self.eps_h = float(self.hyper_step_size)


def dual_averaging_update(st: DualAveragingState, epoch: int, log_accept_ratio: float,
                          dtype=np.float32):
    """network.py:457-469; epoch = iter_ (0-based), m = epoch+1."""
    dt = dtype
    m = dt(epoch) + dt(1)
    lar = dt(log_accept_ratio)
    accept = np.exp(lar) if lar < 0 else dt(1)                       # :459-460
    h = (dt(1) - dt(1) / (m + dt(st.t0))) * dt(st.h) + (dt(1) / (m + dt(st.t0))) * (dt(st.target) - accept)
    log_eps = dt(st.mu) - h * (m ** dt(0.5)) / dt(st.gamma)          # :463
    leb = (dt(1) - m ** (-dt(st.kappa))) * dt(st.log_eps_bar)        # :465
    leb = leb + m ** (-dt(st.kappa)) * log_eps                       # :466
    if m < dt(st.burnin * 0.8):                                      # :468 (Q6)
        st.eps_h = float(np.exp(leb))
    st.h, st.log_eps_bar = float(h), float(leb)
    return float(accept)


# ----------------------------------------------------------------------------
# A16: paramAdapter (tensorBNN/paramAdapter.py), float32 like the reference
# ----------------------------------------------------------------------------
class ParamAdapter:
    """GP-UCB adapter for (eps, L).  ``uniforms`` / ``choices`` let a test
    inject the U(0,1) draws (:232) and the random grid picks (:283-284:
    python ``random.choice``) so a trace is reproducible; when exhausted or
    None, python ``random`` / numpy are used."""

    def __init__(self, e1, L1, el, eu, eNumber, Ll, Lu, lStep, m, k, a=4, delta=0.1,
                 cores=4, strikes=10, randomSteps=10, uniforms=None, choices=None):
        f32 = np.float32
        self.currentE, self.currentL = f32(e1), f32(L1)              # :61-62
        self.el, self.eu = f32(el), f32(eu)
        self.Ll, self.Lu = f32(Ll), f32(Lu)
        self.eNumber = int(eNumber)
        self.eGrid = np.linspace(f32(el), f32(eu), int(eNumber)).astype(f32)   # :68
        self.lGrid = np.array(range(int(Ll), int(Lu) + 1, int(lStep)), dtype=f32)  # :69
        self.lNumber = len(self.lGrid)
        self.delta = f32(delta)
        kappa = f32(0.2)
        self.sigma = np.diag([f32(1) / ((kappa * f32(2)) ** 2)] * 2).astype(f32)  # :72-74
        self.k = k
        self.m = m
        self.a = f32(a)
        self.maxStrikes = 50                                          # :92 (ignores `strikes`)
        self.randomSteps = randomSteps
        self._uniforms = list(uniforms) if uniforms is not None else None
        self._choices = list(choices) if choices is not None else None
        self.sjd_log: List[float] = []
        self._reset_state()
        self.strikes = 0

    def _reset_state(self):
        """reset(), paramAdapter.py:143-156."""
        self.previousGamma = []
        self.allSD = []
        self.K = np.zeros((0, 0), dtype=np.float32)
        self.currentData = []
        self.allData = []
        self.maxR = np.float32(1e-8)
        self.i = -2                                                   # :85
        self.previous_state = None
        self.current_state = None
        self.strikes = 0

    def calck(self, gI, gJ, el=None, eu=None):
        """:95-111 -- exp(-1/2 g1^T Sigma g2): a dot-product form (Q8)."""
        f32 = np.float32
        el = self.el if el is None else el
        eu = self.eu if eu is None else eu
        g1 = np.array([f32(-1) + f32(2) * (f32(gI[0]) - el) / (eu - el),
                       f32(-1) + f32(2) * (f32(gI[1]) - self.Ll) / (self.Lu - self.Ll)], dtype=f32)
        g2 = np.array([f32(-1) + f32(2) * (f32(gJ[0]) - el) / (eu - el),
                       f32(-1) + f32(2) * (f32(gJ[1]) - self.Ll) / (self.Lu - self.Ll)], dtype=f32)
        return f32(np.exp(f32(-0.5) * f32(g1 @ (self.sigma @ g2))))

    def calcUCB(self, test):
        """:113-141 -- ucb = mean + variance * p * rootbeta (Q9)."""
        f32 = np.float32
        kv = np.array([self.calck(g, test) for g in self.previousGamma], dtype=f32)
        mean = f32(kv @ self.inverseR) * self.s
        var = self.calck(test, test) - f32(kv @ (self.inverse @ kv))
        return f32(mean + var * f32(self.p) * self.rootbeta), mean, var

    def gridSearch(self):
        """:158-196 -- e fastest, keep first strictly-greater ucb, init -1e9."""
        f32 = np.float32
        # vectorised but order-preserving (np.argmax returns the first maximum)
        E, Lg = self.eGrid, self.lGrid
        def scale(g):
            return (f32(-1) + f32(2) * (g[:, 0] - self.el) / (self.eu - self.el),
                    f32(-1) + f32(2) * (g[:, 1] - self.Ll) / (self.Lu - self.Ll))
        prev = np.array([[f32(e), f32(l)] for e, l in self.previousGamma], dtype=f32)
        p0, p1 = scale(prev)
        te = (f32(-1) + f32(2) * (E - self.el) / (self.eu - self.el)).astype(f32)
        tl = (f32(-1) + f32(2) * (Lg - self.Ll) / (self.Lu - self.Ll)).astype(f32)
        s0, s1 = self.sigma[0, 0], self.sigma[1, 1]
        best = (f32(-1e9), self.el, self.Ll)
        for li in range(len(Lg)):
            # k[h, e] = exp(-.5*(p0[h]*s0*te[e] + p1[h]*s1*tl[li]))
            kk = np.exp(f32(-0.5) * (np.outer(p0 * s0, te) + (p1 * s1 * tl[li])[:, None])).astype(f32)
            mean = (self.inverseR.astype(f32) @ kk) * self.s
            var = np.exp(f32(-0.5) * (te * s0 * te + tl[li] * s1 * tl[li])).astype(f32) \
                - np.einsum("he,hg,ge->e", kk, self.inverse.astype(f32), kk).astype(f32)
            ucb = (mean + var * f32(self.p) * self.rootbeta).astype(f32)
            j = int(np.argmax(ucb))
            if ucb[j] > best[0]:
                best = (ucb[j], E[j], Lg[li])
        return f32(best[1]), f32(best[2])

    def _uniform(self):
        if self._uniforms:
            return np.float32(self._uniforms.pop(0))
        return np.float32(_pyrandom.random())

    def _choice(self, grid):
        if self._choices:
            return grid[int(self._choices.pop(0)) % len(grid)]
        return _pyrandom.choice(list(grid))

    def update(self, state: Sequence[np.ndarray]):
        """:199-292.  ``state`` is the list of state tensors (or one flat vector)."""
        f32 = np.float32
        if self.i < self.k - 2 and self.strikes == self.maxStrikes:   # :208-214
            self.el = self.el / f32(2)
            self.eu = self.eu / f32(2)
            self.eGrid = np.linspace(self.el, self.eu, self.eNumber).astype(f32)
            self.k = self.k - self.i - 2
            self._reset_state()
            self.strikes = 0
        if isinstance(state, np.ndarray):
            state = [state]
        self.previous_state, self.current_state = self.current_state, [np.array(s, dtype=f32) for s in state]
        if self.previous_state is not None:                           # :218-228
            val = f32(0)
            for old, new in zip(self.previous_state, self.current_state):
                val = f32(val + np.sum(np.square(new.reshape(-1) - old.reshape(-1)), dtype=f32)
                          / f32(self.currentL) ** f32(0.5))
            self.sjd_log.append(float(val))
            self.currentData.append(val)
            if val < 1e-8 and self.i // self.m > self.randomSteps:
                self.strikes += 1
            else:
                self.strikes = 0
        if self.i % self.m == 0 and self.i > 0:                       # :231
            u = self._uniform()                                       # :232
            self.p = max(self.i / self.m - self.k + 1, 1) ** (-0.5)   # :233
            if u < self.p:
                mean = f32(np.mean(np.array(self.currentData, dtype=f32), dtype=f32))
                sd = f32(np.std(np.array(self.currentData, dtype=f32), dtype=f32))
                self.currentData = []
                self.allData.append(mean)
                self.allSD.append(sd)
                self.maxR = f32(np.max(self.allData))
                self.previousGamma.append((self.currentE, self.currentL))   # :242
                size = len(self.previousGamma)
                extra = np.array([self.calck(g, self.previousGamma[-1]) for g in self.previousGamma], dtype=f32)
                K = np.zeros((self.K.shape[0] + 1, self.K.shape[0] + 1), dtype=f32)
                K[:-1, :-1] = self.K
                K[-1, :] = extra
                K[:, -1] = extra
                self.K = K                                            # :244-257
                self.s = f32(self.a / self.maxR)                      # :258
                sigmaNu = f32(np.mean(np.array(self.allSD, dtype=f32), dtype=f32))
                eye = np.eye(K.shape[0], dtype=f32)
                try:                                                  # :263-269
                    self.inverse = np.linalg.inv(K + (sigmaNu ** 2) * eye).astype(f32)
                except np.linalg.LinAlgError:
                    self.inverse = np.linalg.inv(K + (sigmaNu ** 2) * eye + f32(0.1) * eye).astype(f32)
                self.inverseR = (self.inverse @ np.array(self.allData, dtype=f32)).astype(f32)   # :270
                rb = (self.i / self.m + 1) ** 3 * math.pi ** 2        # :274
                rb = f32(rb) / (f32(3) * self.delta)                  # :275
                rb = np.log(f32(rb)) * f32(2)                         # :276
                self.rootbeta = f32(rb ** f32(0.5))                   # :277
                if self.i // self.m >= self.randomSteps:              # :280-284
                    self.currentE, self.currentL = self.gridSearch()
                else:
                    self.currentE = f32(self._choice(self.eGrid))
                    self.currentL = f32(self._choice(self.lGrid))
                if size == 50:                                        # :285-289
                    self.K = self.K[1:, 1:]
                    self.previousGamma = self.previousGamma[1:]
                    self.allData = self.allData[1:]
                    self.allSD = self.allSD[1:]
        self.i += 1
        return np.float32(self.currentE), np.int32(self.currentL)     # :292


# ----------------------------------------------------------------------------
# Chain RNG used by the product when p0 / log u are not injected
# (new -- TF's stream is not reproducible).
# Philox4x32-10, key = (seed, chain_id), counter = (block, epoch, purpose, 0).
# ----------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
PURPOSE_MOMENTUM, PURPOSE_LOGU, PURPOSE_HYPER_MOMENTUM, PURPOSE_HYPER_LOGU = 0, 1, 2, 3


def philox4x32_10(counter, key):
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = PHILOX_M0 * c[0]
        p1 = PHILOX_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + PHILOX_W0) & 0xFFFFFFFF, (k[1] + PHILOX_W1) & 0xFFFFFFFF]
    return c


def _u01(x):
    """uint32 -> float32 in (0,1): ((x>>8)+0.5) * 2^-24."""
    return np.float32((np.float32(x >> 8) + np.float32(0.5)) * np.float32(2.0 ** -24))


def philox_normals(n, seed, chain_id, epoch, purpose):
    """n standard normals: block b = j//4 gives 4 uniforms -> 2 Box-Muller pairs."""
    out = np.empty(((n + 3) // 4) * 4, dtype=np.float32)
    two_pi = np.float32(2 * math.pi)
    for b in range((n + 3) // 4):
        r = philox4x32_10((b, epoch, purpose, 0), (seed, chain_id))
        u = [_u01(x) for x in r]
        for h in range(2):
            rad = np.sqrt(np.float32(-2) * np.log(u[2 * h]))
            out[4 * b + 2 * h] = rad * np.cos(two_pi * u[2 * h + 1])
            out[4 * b + 2 * h + 1] = rad * np.sin(two_pi * u[2 * h + 1])
    return out[:n]


def philox_log_uniform(seed, chain_id, epoch, purpose):
    r = philox4x32_10((0, epoch, purpose, 0), (seed, chain_id))
    return np.float32(np.log(_u01(r[0])))


# ----------------------------------------------------------------------------
# A17 file format (writer network.py:545-559, :610-663; reader predictor.py:43-113)
# ----------------------------------------------------------------------------
class SampleWriter:
    """Restates the file handling of network.train for one chain."""

    def __init__(self, folder, shapes, layer_names, n_hyper, burnin, sampling_step, networks_per_file):
        self.folder = folder
        self.shapes = shapes
        self.n_hyper = n_hyper
        self.burnin, self.sampling_step, self.npf = burnin, sampling_step, networks_per_file
        os.makedirs(folder, exist_ok=True)
        self.files = [open(os.path.join(folder, f"{n}.0.txt"), "wb") for n in range(len(shapes))]   # :552-554
        self.files.append(open(os.path.join(folder, "hypers0.txt"), "wb"))                          # :555
        with open(os.path.join(folder, "architecture.txt"), "wb") as f:                            # :557-559
            for name in layer_names:
                f.write((name + "\n").encode("utf-8"))

    def after_epoch(self, iter_, states, hypers):
        """iter_ is the 1-based epoch count after the increment at network.py:591."""
        shift = iter_ - self.burnin - 1
        interval = self.npf * self.sampling_step
        if iter_ > self.burnin and shift % interval == 0:             # :610-612
            for f in self.files:
                f.close()
            idx = int((iter_ - self.burnin) // interval)
            self.files = [open(os.path.join(self.folder, f"{n}.{idx}.txt"), "wb") for n in range(len(self.shapes))]
            self.files.append(open(os.path.join(self.folder, f"hypers{idx}.txt"), "wb"))
            with open(os.path.join(self.folder, "summary.txt"), "wb") as s:   # :629-646
                for shp in self.shapes:
                    s.write((" ".join(str(x) for x in shp).strip() + "\n").encode("utf-8"))
                num_networks = shift // self.sampling_step
                num_files = num_networks // self.npf + (1 if num_networks % self.npf else 0)
                s.write(f"{num_networks} {num_files} {len(self.shapes)}\n".encode("utf-8"))
                s.write(str(self.n_hyper).encode("utf-8"))
        if iter_ > self.burnin and iter_ % self.sampling_step == 0:   # :648-663
            for f, st in zip(self.files[:-1], states):
                np.savetxt(f, st)
            np.savetxt(self.files[-1], [np.reshape(h, (1,)) for h in hypers])

    def close(self):
        for f in self.files:
            f.close()


def load_networks(folder):
    """predictor.loadNetworks, predictor.py:43-113 (returns matrices, hypers)."""
    summary = []
    with open(os.path.join(folder, "summary.txt"), "r") as file:
        for line in file:
            summary.append(line.split())
    num_networks = int(summary[-2][0])
    num_files = int(summary[-2][1])
    num_matrices = int(summary[-2][2])
    num_hypers = int(summary[-1][0])
    num_networks //= num_files
    matrices = []
    for n in range(num_matrices):
        d1 = int(summary[n][0])
        d2 = int(summary[n][1]) if len(summary[n]) == 2 else 1
        w0 = np.zeros((num_networks * num_files, d1, d2), dtype=np.float32)
        for m in range(num_files):
            w = np.loadtxt(os.path.join(folder, f"{n}.{m}.txt"), dtype=np.float32, ndmin=2)
            for k in range(num_networks):
                w0[m * num_networks + k] = w[d1 * k:d1 * (k + 1), :d2]
        matrices.append(w0)
    hypers = []
    if num_hypers > 0:
        for m in range(num_files):
            w = np.loadtxt(os.path.join(folder, f"hypers{m}.txt"), dtype=np.float32, ndmin=1)
            for k in range(num_networks):
                hypers.append(w[num_hypers * k:num_hypers * (k + 1)])
    return matrices, hypers


# ----------------------------------------------------------------------------
# Synthetic workloads of BASELINE.md section 3 / SURVEY section 8(d)
# ----------------------------------------------------------------------------
def make_spec(dims, act=ACT_RELU, prior=PRIOR_CAUCHY, likelihood=LIK_GAUSSIAN, final_act=ACT_NONE,
              fixed_sd=0.1):
    layers = []
    for i in range(len(dims) - 1):
        last = i == len(dims) - 2
        layers.append(DenseSpec(dims[i], dims[i + 1], final_act if last else act, prior))
    return NetSpec(layers, likelihood, fixed_sd)


def synth_problem(dims, n, act=ACT_RELU, prior=PRIOR_CAUCHY, likelihood=LIK_GAUSSIAN):
    """X~N(0,1) PCG64(1234); teacher weights N(0,sqrt(2/out)) PCG64(4321);
    regression: Y = teacher(X)+N(0,.1^2) PCG64(5678), standardised;
    classification: Y~Bernoulli(sigmoid(teacher logits)).  Initial chain state
    N(0,sqrt(2/out)) (layer.py:253-262) from PCG64(1000*(layer+1)) (+1 for biases)."""
    final_act = ACT_SIGMOID if likelihood == LIK_BERNOULLI else ACT_NONE
    spec = make_spec(dims, act, prior, likelihood, final_act)
    X = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dims[0])).astype(np.float32)
    tg = np.random.Generator(np.random.PCG64(4321))
    tparts = []
    for l in spec.layers:
        sd = (2.0 / l.out_dim) ** 0.5
        tparts.append(((tg.standard_normal((l.out_dim, l.in_dim)) * sd).astype(np.float32),
                       (tg.standard_normal((l.out_dim, 1)) * sd).astype(np.float32)))
    # teacher outputs and targets in fp64, rounded to fp32 ONCE: an fp32 matmul's last bits follow the BLAS thread count, and
    # inputs that do must not decide a tolerance (fp64 sums differ by 1e-16 relative at most: invisible after the rounding)
    f = forward(spec, flatten(tparts).astype(np.float64), X.astype(np.float64), np.float64)
    ng = np.random.Generator(np.random.PCG64(5678))
    if likelihood == LIK_BERNOULLI:
        Y = (ng.random(f.T.shape) < f.T).astype(np.float32)
    else:
        Y = f.T + 0.1 * ng.standard_normal(f.T.shape).astype(np.float32).astype(np.float64)
        sd_y = Y.std(0)
        Y = ((Y - Y.mean(0)) / np.where(sd_y > 0, sd_y, 1.0)).astype(np.float32)
    parts = []
    for i, l in enumerate(spec.layers):
        sd = (2.0 / l.out_dim) ** 0.5
        W = (np.random.Generator(np.random.PCG64(1000 * (i + 1))).standard_normal((l.out_dim, l.in_dim)) * sd)
        b = (np.random.Generator(np.random.PCG64(1000 * (i + 1) + 1)).standard_normal((l.out_dim, 1)) * sd)
        parts.append((W.astype(np.float32), b.astype(np.float32)))
    theta0 = flatten(parts).astype(np.float32)
    eta0 = default_hypers(spec, 0.1)
    return spec, X, Y, theta0, eta0
