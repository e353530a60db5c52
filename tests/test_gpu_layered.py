"""GPU: the layered MFMA family (kernels_layered.hpp) -- what an architecture runs on when no shape-specialised fused kernel
covers it: fan-in above 32 (the reference's MNIST example, docs/ClassificationExample.md: 784 -> 20 -> 20 -> 1), mixed hidden
activations, more than two outputs, widths above 256.  Value, gradient and statistic against the fp64 oracle and the
thread-per-row kernel; transitions against the oracle; forward against the oracle."""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

LOGP_RTOL = 4e-6


def scaled_problem(dims, n, acts, prior, lik, seed=0):
    """synth_problem's recipe with per-layer activations and inputs scaled so that wide fan-ins do not saturate"""
    final = o.ACT_SIGMOID if lik == o.LIK_BERNOULLI else o.ACT_NONE
    spec = o.make_spec(dims, acts[0], prior, lik, final)
    for l, a in zip(spec.layers[:-1], acts):
        l.act = a
    rng = np.random.Generator(np.random.PCG64(100 + seed))
    X = (rng.standard_normal((n, dims[0])) / np.sqrt(dims[0])).astype(np.float32)
    parts = []
    for l in spec.layers:
        sd = (2.0 / l.out_dim) ** 0.5
        parts.append(((rng.standard_normal((l.out_dim, l.in_dim)) * sd).astype(np.float32),
                      (rng.standard_normal((l.out_dim, 1)) * sd).astype(np.float32)))
    theta = o.flatten(parts).astype(np.float32)
    f = o.forward(spec, theta, X, np.float64).T
    if lik == o.LIK_BERNOULLI:
        Y = (rng.random(f.shape) < f).astype(np.float32)
    else:
        Y = (f + 0.3 * rng.standard_normal(f.shape)).astype(np.float32)
    theta0 = (theta + 0.1 * rng.standard_normal(theta.shape)).astype(np.float32)
    eta = o.default_hypers(spec, 0.1)
    return spec, X, Y, theta0, eta


CASES = {
    "mnist_like": ([784, 20, 20, 1], 1500, [o.ACT_RELU, o.ACT_RELU], o.PRIOR_CAUCHY, o.LIK_BERNOULLI),     # split-K GEMMs
    # 784 -> 100 -> 100 -> 10 (network.add takes any stack, network.py:173-191): 10 outputs on the tile likelihood; at 6,000 rows the first
    # layer's GEMM has 1,500 (row tile, tile group) items: two row tiles per item AND the k-groups split over the workgroup (round 5)
    "ten_class": ([784, 100, 100, 10], 6000 + 3, [o.ACT_RELU, o.ACT_RELU], o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
    "tabular100": ([100, 50, 50, 1], 3000 + 5, [o.ACT_RELU, o.ACT_RELU], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "tabular100_big": ([100, 50, 50, 1], 70000 + 9, [o.ACT_RELU, o.ACT_RELU], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),   # two row tiles per wave
    "mixed_acts": ([4, 8, 8, 1], 777, [o.ACT_RELU, o.ACT_TANH], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "three_out": ([6, 33, 17, 3], 1000, [o.ACT_TANH, o.ACT_SIGMOID], o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "wide300": ([8, 300, 300, 1], 640, [o.ACT_RELU, o.ACT_RELU], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "exact16": ([16, 32, 16, 2], 333, [o.ACT_ELU, o.ACT_ELU], o.PRIOR_CAUCHY, o.LIK_BERNOULLI),     # ones slots open a tile of their own
    "many_out": ([9, 30, 20], 450, [o.ACT_TANH], o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),                  # 20 outputs: two tiles in the likelihood
    "one_layer": ([40, 1], 500, [], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "deep": ([5, 12, 12, 12, 12, 12, 12, 1], 900, [o.ACT_TANH] * 6, o.PRIOR_CAUCHY, o.LIK_FIXED_GAUSSIAN),
}


@pytest.fixture(autouse=True)
def layered_family_only(monkeypatch):
    """this module tests the layered family: the 784 -> 20 -> 20 -> 1 cases stay on it although the tall-fan-in fused kernel
    (kernels_tall.hpp, tests/test_gpu_tall.py) is what a chain gets for that shape by default"""
    monkeypatch.setenv("TBNN_TALL", "0")
    monkeypatch.setenv("TBNN_REGISTERED", "0")   # nor on a kernel library another test module registered in this process


def make_chain(native, spec, kernel, **kw):
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=kernel, jit=False, **kw)


@pytest.mark.parametrize("case", list(CASES))
def test_logp_grad_layered(native, case):
    dims, n, acts, prior, lik = CASES[case]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts or [o.ACT_RELU], prior, lik)
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    assert ch.kernel_name.startswith("layered<"), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp2, g2, st2 = ch.logp_grad(theta, eta)
    assert lp == lp2 and st == st2 and np.array_equal(g, g2)                  # deterministic
    ch.close()
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (case, a, b)
    gen = make_chain(native, spec, native.KERNEL_GENERIC)
    gen.set_data(X, Y)
    lpg, gg, stg = gen.logp_grad(theta, eta)
    gen.close()
    assert abs(st - stg) <= 2e-6 * abs(stg) + 1e-4


@pytest.mark.parametrize("n", [1, 15, 16, 17, 129, 2048 + 3])
def test_logp_grad_layered_ragged_rows(native, n):
    spec, X, Y, theta, eta = scaled_problem([40, 24, 3], n, [o.ACT_TANH], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    assert ch.kernel_name.startswith("layered<")
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    ch.close()
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
    assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()


@pytest.mark.parametrize("case", ["mnist_like", "mixed_acts", "three_out"])
def test_transition_layered(native, case):
    dims, n, acts, prior, lik = CASES[case]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts, prior, lik)
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    eps, L = 2e-3, 5
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    out = ch.hmc_step(eps, L, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, theta, eta, X, Y, eps, L, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio)
    assert bool(out["accepted"]) == ref.accepted
    np.testing.assert_allclose(ch.get_state(), ref.theta, rtol=2e-4, atol=2e-5)
    # a free-running stretch (no trace: the image k_update maintains is the one the GEMMs read)
    outs = ch.hmc_run(eps, L, 4)
    assert all(np.isfinite(o_["log_accept_ratio"]) for o_ in outs)
    ch.close()


@pytest.mark.parametrize("case", ["mnist_like", "three_out", "exact16"])
def test_forward_layered(native, case):
    dims, n, acts, prior, lik = CASES[case]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts, prior, lik)
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    f = ch.forward(X[:301], theta)
    ch.close()
    ref = o.forward(spec, theta, X[:301], np.float64)
    np.testing.assert_allclose(f, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("case", ["mnist_like", "tabular100", "three_out", "deep", "one_layer"])
def test_fused_tail_matches_separate_launches(native, monkeypatch, case):
    """k_lay_tail (the narrow suffix of layers + likelihood + their delta chain in one launch; row counts below 32 k) against one
    launch per layer and direction (TBNN_LAY_TAIL=0): value, statistic and gradient"""
    dims, n, acts, prior, lik = CASES[case]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts or [o.ACT_RELU], prior, lik)
    res = []
    for tail in ("1", "0"):
        monkeypatch.setenv("TBNN_LAY_TAIL", tail)
        ch = make_chain(native, spec, native.KERNEL_AUTO)
        ch.set_data(X, Y)
        res.append(ch.logp_grad(theta, eta))
        ch.close()
    (lp1, g1, st1), (lp0, g0, st0) = res
    assert abs(lp1 - lp0) <= 1e-6 * abs(lp0) and abs(st1 - st0) <= 1e-6 * abs(st0)
    assert np.abs(g1 - g0).max() <= 2e-5 * np.abs(g0).max()


def test_forward_many_layered(native):
    dims, n, acts, prior, lik = CASES["three_out"]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts, prior, lik)
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    thetas = np.stack([theta, theta * 0.5, theta + 0.25]).astype(np.float32)
    out = ch.forward_many(thetas, X[:200])
    ch.set_validation(X[:77], Y[:77])
    out_v = ch.forward_many(thetas, None, which=1)
    ch.close()
    for i in range(3):
        ref = o.forward(spec, thetas[i], X[:200], np.float64)
        np.testing.assert_allclose(out[i], ref, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(out_v[i], ref[:, :77], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("family", ["tall", "layered"])
def test_mnist_shaped_classification_flow(tmp_path, monkeypatch, native, family):
    """docs/ClassificationExample.md's network -- 784 inputs in [0, 1], two hidden layers of 20, one sigmoid output,
    BernoulliLikelihood -- through the drop-in Python API on synthetic 'images': on the fused tall-fan-in kernel (what a user gets)
    and on the layered family (TBNN_TALL=0); learns, writes samples the predictor reads back."""
    monkeypatch.setenv("TBNN_TALL", "1" if family == "tall" else "0")
    from tensorbnn_amd.activationFunctions import Relu, Sigmoid
    from tensorbnn_amd.layer import DenseLayer
    from tensorbnn_amd.likelihood import BernoulliLikelihood
    from tensorbnn_amd.metrics import Accuracy
    from tensorbnn_amd.network import network
    from tensorbnn_amd.predictor import predictor
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("TBNN_JIT", "0")
    rng = np.random.default_rng(11)
    proto = rng.random((2, 784)).astype(np.float32)                       # two 'digits'
    lab = (rng.random(1200) < 0.5).astype(np.float32)
    X = np.clip(proto[lab.astype(int)] * 0.6 + 0.4 * rng.random((1200, 784)), 0, 1).astype(np.float32)
    net = network(np.float32, 784, X[:1000], lab[:1000, None], X[1000:], lab[1000:, None])
    net.add(DenseLayer(784, 20, seed=1000)); net.add(Relu())
    net.add(DenseLayer(20, 20, seed=2000)); net.add(Relu())
    net.add(DenseLayer(20, 1, seed=3000)); net.add(Sigmoid())
    net.setupMCMC(stepSizeStart=1e-3, stepSizeMin=2e-4, stepSizeMax=4e-3, stepSizeOptions=10, leapfrogStart=20, leapfogMin=10,
                  leapFrogMax=40, leapfrogIncrement=5, hyperStepSize=1e-4, hyperLeapfrog=10, burnin=20, averagingSteps=5)
    rec = net.train(50, 5, BernoulliLikelihood(), metricList=[Accuracy()], adjustHypers=True, folderName="mnist", networksPerFile=2,
                    displaySkip=25)
    assert net._chain.kernel_name == ("tall<relu,sigmoid,bernoulli;784,20,20,1>" if family == "tall" else "layered<784,20,20,1>")
    assert len(rec) == 50 and np.mean([r["main"]["accept_prob"] for r in rec]) > 0.2
    p = predictor(str(tmp_path / "mnist") + "/", likelihood=BernoulliLikelihood())
    preds = np.array(p.predict(X[1000:]))
    assert preds.shape[1:] == (1, 200)
    assert np.mean((preds.mean(axis=0)[0] > 0.5) == (lab[1000:] > 0.5)) > 0.9
    spec = o.make_spec([784, 20, 20, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, final_act=o.ACT_SIGMOID)
    np.testing.assert_allclose(preds[0], o.forward(spec, p.vectors[0], X[1000:], np.float64), rtol=5e-5, atol=2e-5)


def test_layered_data_swap_metrics_and_hyper_step(native):
    """the same chain over two data sets of different size (the activation store is rebuilt), the device metrics over the
    staged rows, and a hyper transition followed by a weight transition (the O(P) refresh of the cached gradient)"""
    dims, n, acts, prior, lik = CASES["three_out"]
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts, prior, lik)
    ch = make_chain(native, spec, native.KERNEL_AUTO, seed=3)
    for rows in (n, 171, 640):
        ch.set_data(X[:rows], Y[:rows])
        lp, g, st = ch.logp_grad(theta, eta)
        lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X[:rows], Y[:rows], np.float64)[:2]
        assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
        assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()
    ch.set_validation(X[:200], Y[:200])
    ch.set_state(theta); ch.set_hypers(eta)
    m = ch.metrics(which=1)
    f = o.forward(spec, theta, X[:200], np.float64).T
    assert abs(m[0] - np.mean((f - Y[:200]) ** 2)) <= 1e-5 * max(1.0, m[0])
    # hyper transition, then a weight transition from the refreshed cache against a fresh evaluation at the new hypers
    p0h = np.random.default_rng(1).standard_normal(spec.n_hypers).astype(np.float32)
    ch.hmc_step(1e-3, 2)
    hs = ch.hyper_step(1e-3, 5, p0=p0h, log_u=-1e30)
    assert hs["accepted"] == 1
    th, et = ch.get_state(), ch.get_hypers()
    p0 = np.random.default_rng(2).standard_normal(spec.n_params).astype(np.float32)
    out = ch.hmc_step(1e-3, 3, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, th, et, X[:640], Y[:640], 1e-3, 3, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    ch.close()


@pytest.mark.parametrize("seed", range(12))
def test_layered_random_architectures(native, monkeypatch, seed):
    """random depth / widths / fan-in / output count / activations / likelihood / row count (tile-boundary widths 15-17, 31-33
    favoured): value and gradient against the fp64 oracle, with and without the fused tail"""
    rng = np.random.default_rng(1000 + seed)
    edge = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65]
    pick = lambda hi: int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, hi))
    nl = int(rng.integers(1, 6))
    dims = [pick(90)] + [pick(70) for _ in range(nl - 1)] + [int(rng.integers(1, 6))]
    acts = [int(rng.choice([o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU])) for _ in range(nl - 1)]
    lik = int(rng.choice([o.LIK_GAUSSIAN, o.LIK_BERNOULLI, o.LIK_FIXED_GAUSSIAN]))
    prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    n = int(rng.choice([1, 16, 17, 100, 333, 1000]))
    spec, X, Y, theta, eta = scaled_problem(dims, n, acts or [o.ACT_RELU], prior, lik, seed=seed)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    for tail in ("1", "0"):
        monkeypatch.setenv("TBNN_LAY_TAIL", tail)
        monkeypatch.setenv("TBNN_JIT", "0")
        layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
        ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_AUTO, jit=False)
        if not ch.kernel_name.startswith("layered<"):          # a registry shape by chance
            ch.close()
            pytest.skip("registry shape")
        ch.set_data(X, Y)
        lp, g, st = ch.logp_grad(theta, eta)
        ch.close()
        assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3, (dims, acts, lik, n, tail)
        for l, (ow, ob) in zip(spec.layers, spec.offsets()):
            for a, b in ((ow, ob), (ob, ob + l.out_dim)):
                assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (dims, acts, lik, n, tail, a, b)
