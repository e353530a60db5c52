"""Synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8(d).

Data generation only (NumPy): X ~ N(0,1) PCG64(1234); teacher network of the
same architecture with weights N(0, sqrt(2/out)) PCG64(4321); regression
Y = teacher(X) + N(0, 0.1^2) PCG64(5678), standardised; classification
Y ~ Bernoulli(sigmoid(teacher logits)).  Initial chain state N(0, sqrt(2/out))
(reference layer.py:253-262) from PCG64(1000*(layer+1)) (+1 for biases).
"""
import os

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


# BASELINE.json configs as workloads: c1 = configs[0] (plumbing), c2 = configs[1] (the metric's config; configs[2] is c2
# on every GPU), c4 = configs[3], c5 = configs[4].  steps / warmup: bench.py's defaults per workload.
WORKLOADS = {
    "c1": dict(dims=[1, 10, 10, 1], n=1_000, L=100, lik=nat.LIK_GAUSSIAN, hyper=False, steps=100, warmup=10,
               text="BASELINE configs[0]: 1->10->10->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood sd=0.1), "
                    "1k-row fp32 synthetic regression, L=100 leapfrog, 1 chain (launch-latency-bound plumbing case)"),
    "c2": dict(dims=[5, 50, 50, 50, 1], n=100_000, L=50, lik=nat.LIK_GAUSSIAN, hyper=False, steps=200, warmup=20,
               text="BASELINE configs[1]: 5->50->50->50->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood sd=0.1), "
                    "100k-row fp32 synthetic regression, L=50 leapfrog, 1 chain per GPU"),
    "c4": dict(dims=[10, 200, 200, 200, 1], n=1_000_000, L=100, lik=nat.LIK_GAUSSIAN, hyper=False, steps=10, warmup=2,
               text="BASELINE configs[3]: 10->200->200->200->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood sd=0.1), "
                    "1M-row fp32 synthetic regression, L=100 leapfrog, 1 chain per GPU"),
    # not a BASELINE config: the reference's own classification tutorial (docs/ClassificationExample.md:103-173: 784 -> 20 -> 20 -> 1 on
    # MNIST pixels, two digits = 12k rows) with pixel-like synthetic rows |N(0,1)| / 28 -- the shape the tall-fan-in fused family is for
    "mn": dict(dims=[784, 20, 20, 1], n=12_000, L=50, lik=nat.LIK_BERNOULLI, hyper=False, steps=40, warmup=5, x_scale=1.0 / 28.0,
               text="docs/ClassificationExample.md shape: 784->20->20->1 Relu/Sigmoid BNN (Cauchy DenseLayer, BernoulliLikelihood), "
                    "12k pixel-like fp32 synthetic rows, L=50 leapfrog, 1 chain per GPU"),
    "c5": dict(dims=[20, 100, 100, 2], n=500_000, L=50, lik=nat.LIK_BERNOULLI, hyper=True, steps=40, warmup=5,
               text="BASELINE configs[4]: 20->100->100->2 Relu/Sigmoid BNN (Cauchy DenseLayer, BernoulliLikelihood), "
                    "500k-row fp32 synthetic classification, L=50 leapfrog + hyper-HMC (L_h=100, dual averaging) per epoch, "
                    "1 chain per GPU"),
    # configs[4] with the reference's other prior family: under the Cauchy layers' quirk Q1 the hyper target is improper
    # (it grows without bound as a scale g^2 -> 0) and the hyper chain collapses onto that pole whatever the sampler;
    # GaussianDenseLayer priors (layer.py:282-459) give the hyper transition a proper target
    "c5g": dict(dims=[20, 100, 100, 2], n=500_000, L=50, lik=nat.LIK_BERNOULLI, hyper=True, steps=40, warmup=5, prior=nat.PRIOR_GAUSSIAN,
                text="BASELINE configs[4] with GaussianDenseLayer priors: 20->100->100->2 Relu/Sigmoid BNN, BernoulliLikelihood, "
                     "500k-row fp32 synthetic classification, L=50 leapfrog + hyper-HMC (L_h=100, dual averaging) per epoch, "
                     "1 chain per GPU"),
    # not BASELINE configs: two architectures outside the shape-specialised families' first reach, profiled every round
    # (tools/profile_round.sh) -- wide hidden layers, and the 10-class net of network.add's "any stack" (network.py:173-191)
    "w300": dict(dims=[8, 300, 300, 1], n=50_000, L=20, lik=nat.LIK_GAUSSIAN, hyper=False, steps=10, warmup=2,
                 text="8->300->300->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood), 50k-row fp32 synthetic regression, L=20 leapfrog, 1 chain per GPU"),
    "mc10": dict(dims=[784, 100, 100, 10], n=12_000, L=20, lik=nat.LIK_BERNOULLI, hyper=False, steps=10, warmup=2, x_scale=1.0 / 28.0,
                 text="784->100->100->10 Relu/Sigmoid BNN (Cauchy DenseLayer, BernoulliLikelihood over 10 outputs), 12k pixel-like fp32 synthetic rows, "
                      "L=20 leapfrog, 1 chain per GPU"),
    # round 6: a network with MORE THAN TWO outputs on the wide family (its last layer one more streamed middle layer): configs[3]'s widths, ten classes
    "wm10": dict(dims=[10, 200, 200, 10], n=100_000, L=20, lik=nat.LIK_BERNOULLI, hyper=False, steps=10, warmup=2,
                 text="10->200->200->10 Relu/Sigmoid BNN (Cauchy DenseLayer, BernoulliLikelihood over 10 outputs), 100k-row fp32 synthetic classification, "
                      "L=20 leapfrog, 1 chain per GPU"),
    # late round 6: two shapes that reach a fused kernel through limits of jit.py that were relaxed: the canonical one-hidden-layer demo on
    # the narrow kernels, and the wide family behind 50 inputs
    "oh100": dict(dims=[1, 100, 1], n=100_000, L=20, lik=nat.LIK_GAUSSIAN, hyper=False, steps=10, warmup=2,
                  text="1->100->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood), 100k-row fp32 synthetic regression, L=20 leapfrog, 1 chain per GPU"),
    "wf50": dict(dims=[50, 100, 100, 1], n=100_000, L=20, lik=nat.LIK_GAUSSIAN, hyper=False, steps=10, warmup=2,
                 text="50->100->100->1 Relu BNN (Cauchy DenseLayer, GaussianLikelihood), 100k-row fp32 synthetic regression, L=20 leapfrog, 1 chain per GPU"),
}
for _w in WORKLOADS.values():
    _w.setdefault("prior", nat.PRIOR_CAUCHY)


def burned_state(cfg: str, directory: str):
    """a burned-in chain state of a config written by tools/make_burned.py into `directory` (the caller says where its fixtures
    live: bench.py passes tests/golden), or None: dict with theta, eta, eps, L, accept and, for configs[4], the dual-averaging
    state of the hyper step size"""
    path = os.path.join(directory, f"{cfg}_burned.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return {k: z[k] for k in z.files}


def _act(z, a):
    if a == nat.ACT_RELU:
        return np.maximum(z, 0)
    if a == nat.ACT_TANH:
        return np.tanh(z)
    if a == nat.ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-z))
    return z


def synth_problem(dims, n, act=nat.ACT_RELU, prior=nat.PRIOR_CAUCHY, likelihood=nat.LIK_GAUSSIAN, x_scale=None):
    """returns (layers, likelihood, X[n,d_in], Y[n,d_out], theta0[P], eta0[H]); x_scale: rows |N(0,1)| * x_scale (pixel-like)
    instead of N(0,1)"""
    nl = len(dims) - 1
    final_act = nat.ACT_SIGMOID if likelihood == nat.LIK_BERNOULLI else nat.ACT_NONE
    layers = [(dims[i], dims[i + 1], final_act if i == nl - 1 else act, prior) for i in range(nl)]
    X = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dims[0])).astype(np.float32)
    if x_scale is not None:
        X = (np.abs(X) * np.float32(x_scale)).astype(np.float32)
    tg = np.random.Generator(np.random.PCG64(4321))
    # (fp32 teacher weights, fp64 arithmetic, targets rounded to fp32 once: an fp32 matmul's last bits follow the BLAS thread count)
    a = X.T.astype(np.float64)
    for (i, o, ac, _) in layers:
        sd = (2.0 / o) ** 0.5
        W = (tg.standard_normal((o, i)) * sd).astype(np.float32).astype(np.float64)
        b = (tg.standard_normal((o, 1)) * sd).astype(np.float32).astype(np.float64)
        a = _act(W @ a + b, ac)
    f = a
    ng = np.random.Generator(np.random.PCG64(5678))
    if likelihood == nat.LIK_BERNOULLI:
        Y = (ng.random(f.T.shape) < f.T).astype(np.float32)
    else:
        Y = f.T + 0.1 * ng.standard_normal(f.T.shape).astype(np.float32).astype(np.float64)
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
