"""Synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8(d).

Data generation only (NumPy): X ~ N(0,1) PCG64(1234); teacher network of the
same architecture with weights N(0, sqrt(2/out)) PCG64(4321); regression
Y = teacher(X) + N(0, 0.1^2) PCG64(5678), standardised; classification
Y ~ Bernoulli(sigmoid(teacher logits)).  Initial chain state N(0, sqrt(2/out))
(reference layer.py:253-262) from PCG64(1000*(layer+1)) (+1 for biases).
"""
import numpy as np

from . import _native as nat

# fixed leapfrog step sizes per config (adapter bypassed for timing): (warm-up eps, timed eps).
# The chain starts far from equilibrium (log-prob -2.7e7), where only eps <= 4e-5 is stable; the
# warm-up epochs use the safe value, the timed epochs the value that puts the mean acceptance
# probability in [0.6, 0.9] (scan recorded in tests/golden/bench_eps.json).
_BENCH_EPS = {"c2": (2.0e-5, 8.0e-5),
              "c4": (1.0e-6, 1.6e-5),      # configs[3], L=100: scan 8e-6 -> 0.93, 1.6e-5 -> 0.86, 3.2e-5 unstable
              "c5": (5.0e-5, 2.0e-4)}      # configs[4], L=50: scan 1e-4 -> 0.99, 2e-4 -> 0.84, 4e-4 -> 0.75


def bench_eps(cfg: str):
    return _BENCH_EPS[cfg]


def _act(z, a):
    if a == nat.ACT_RELU:
        return np.maximum(z, 0)
    if a == nat.ACT_TANH:
        return np.tanh(z)
    if a == nat.ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-z))
    return z


def synth_problem(dims, n, act=nat.ACT_RELU, prior=nat.PRIOR_CAUCHY, likelihood=nat.LIK_GAUSSIAN):
    """returns (layers, likelihood, X[n,d_in], Y[n,d_out], theta0[P], eta0[H])"""
    nl = len(dims) - 1
    final_act = nat.ACT_SIGMOID if likelihood == nat.LIK_BERNOULLI else nat.ACT_NONE
    layers = [(dims[i], dims[i + 1], final_act if i == nl - 1 else act, prior) for i in range(nl)]
    X = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dims[0])).astype(np.float32)
    tg = np.random.Generator(np.random.PCG64(4321))
    a = X.T.astype(np.float32)
    for (i, o, ac, _) in layers:
        sd = (2.0 / o) ** 0.5
        W = (tg.standard_normal((o, i)) * sd).astype(np.float32)
        b = (tg.standard_normal((o, 1)) * sd).astype(np.float32)
        a = _act(W @ a + b, ac).astype(np.float32)
    f = a
    ng = np.random.Generator(np.random.PCG64(5678))
    if likelihood == nat.LIK_BERNOULLI:
        Y = (ng.random(f.T.shape) < f.T).astype(np.float32)
    else:
        Y = f.T + 0.1 * ng.standard_normal(f.T.shape).astype(np.float32)
        sd_y = Y.std(0)
        Y = ((Y - Y.mean(0)) / np.where(sd_y > 0, sd_y, 1.0)).astype(np.float32)
    parts = []
    for li, (i, o, _, _) in enumerate(layers):
        sd = (2.0 / o) ** 0.5
        W = np.random.Generator(np.random.PCG64(1000 * (li + 1))).standard_normal((o, i)) * sd
        b = np.random.Generator(np.random.PCG64(1000 * (li + 1) + 1)).standard_normal((o, 1)) * sd
        parts += [W.astype(np.float32).reshape(-1), b.astype(np.float32).reshape(-1)]
    theta0 = np.concatenate(parts).astype(np.float32)
    eta = []
    for (_, _, _, pr) in layers:
        eta += [0.0, 0.5 ** 0.5, 0.0, 0.5 ** 0.5] if pr == nat.PRIOR_CAUCHY else [0.0, 1.0, 0.0, 1.0]
    if likelihood == nat.LIK_GAUSSIAN:
        eta.append(0.1 ** 0.5)
    return layers, likelihood, X, Y, theta0, np.asarray(eta, dtype=np.float32)
