"""GPU: predictions and metrics over the staged rows (tbnn_predict / tbnn_metrics)."""
import numpy as np
import pytest

import tbnn_oracle as o
from tensorbnn_amd import metrics as M

pytestmark = pytest.mark.gpu

CASES = {
    "narrow": ([5, 50, 50, 50, 1], 2000, o.ACT_RELU, o.LIK_GAUSSIAN),          # k_forward_fast3 (forward-only MFMA kernel)
    "generic": ([3, 7, 5, 2], 900, o.ACT_TANH, o.LIK_GAUSSIAN),                  # k_forward_generic
    "mid": ([20, 100, 100, 2], 1500, o.ACT_RELU, o.LIK_BERNOULLI),             # k_fwd_bwd_mid<S, FWD>
    "mid_two_middle": ([7, 33, 18, 50, 2], 800, o.ACT_RELU, o.LIK_GAUSSIAN),
    "wide_stream": ([10, 200, 200, 200, 1], 1100, o.ACT_RELU, o.LIK_GAUSSIAN), # k_chain_wide<S, FWD>, streamed weights
    "wide_ring_small": ([20, 32, 16, 48, 2], 700, o.ACT_SIGMOID, o.LIK_BERNOULLI),
}


@pytest.mark.parametrize("case", list(CASES))
def test_predict_and_metrics(native, case):
    dims, n, act, lik = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, lik)
    nv = n // 3 + 5
    Xv, Yv = X[:nv] * 0.5 + 0.1, Y[:nv]
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, kernel=native.KERNEL_AUTO)
    ch.set_data(X, Y); ch.set_validation(Xv, Yv); ch.set_state(theta)
    ref_t = o.forward(spec, theta, X, np.float64)
    ref_v = o.forward(spec, theta, Xv, np.float64)
    pt, pv = ch.predict(0), ch.predict(1)
    np.testing.assert_allclose(pt, ref_t, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pv, ref_v, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ch.forward(Xv, theta), ref_v, rtol=2e-5, atol=2e-5)     # tbnn_forward takes the same kernel
    # another theta without touching the chain state
    th2 = (theta * 0.9).astype(np.float32)
    np.testing.assert_allclose(ch.predict(1, th2), o.forward(spec, th2, Xv, np.float64), rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(ch.get_state(), theta)
    # the three reference metrics (host NumPy restatement of metrics.py as the checker), incl. the scaleExp quirk
    for scale_exp, mean, sd in ((False, 0.0, 1.0), (True, 0.3, 0.7)):
        for cls, idx in ((M.SquaredError, 0), (M.PercentError, 1), (M.Accuracy, 2)):
            m = cls(scaleExp=scale_exp, mean=mean, sd=sd)
            m.calculate(pt, pv, Y, Yv)
            pv_exp = scale_exp and cls is not M.SquaredError
            tr = ch.metrics(0, None, mean, sd, scale_exp, scale_exp)[idx]
            va = ch.metrics(1, None, mean, sd, pv_exp, scale_exp)[idx]
            want_t, want_v = {0: (m.squaredErrorTrain, m.squaredErrorValidate) if idx == 0 else None,
                              1: (m.percentErrorTrain, m.percentErrorValidate) if idx == 1 else None,
                              2: (1 - m.accuracyTrain, 1 - m.accuracyValidate) if idx == 2 else None}[idx]
            for got, want in ((tr, want_t), (va, want_v)):
                if np.isfinite(want):
                    assert abs(got - want) <= 2e-4 * abs(want) + 1e-6, (cls.__name__, scale_exp, got, want)
                else:                       # percent error against zero targets: inf (or nan) on both sides
                    assert not np.isfinite(got), (cls.__name__, scale_exp, got, want)
    ch.close()


ENSEMBLE = {
    "narrow_batched": ([5, 50, 50, 50, 1], 1777, o.ACT_RELU, o.LIK_GAUSSIAN, 37),     # one launch, grid.y = network
    "narrow_tanh": ([1, 10, 10, 10, 1], 333, o.ACT_TANH, o.LIK_GAUSSIAN, 5),
    "mid": ([20, 100, 100, 2], 900, o.ACT_RELU, o.LIK_BERNOULLI, 4),                  # one launch of k_fwd_bwd_mid<S, FWD>, grid.y = network
    "wide": ([20, 32, 16, 48, 2], 700, o.ACT_SIGMOID, o.LIK_BERNOULLI, 3),            # per-network k_chain_wide<S, FWD>
    "generic": ([3, 7, 5, 2], 500, o.ACT_SIGMOID, o.LIK_GAUSSIAN, 6),
}


@pytest.mark.parametrize("case", list(ENSEMBLE))
def test_forward_many(native, case):
    """tbnn_forward_many (predictor.predict, predictor.py:132-155): every network of an ensemble against the oracle,
    over host rows and over the staged validation rows; the chain state stays untouched"""
    dims, n, act, lik, m = ENSEMBLE[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, lik)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, kernel=native.KERNEL_AUTO)
    ch.set_data(X, Y); ch.set_state(theta)
    rng = np.random.default_rng(5)
    thetas = (theta[None, :] * (1.0 + 0.3 * rng.standard_normal((m, theta.size)))).astype(np.float32)
    Xq = (X[: n // 2 + 3] * 0.7 - 0.05).astype(np.float32)
    got = ch.forward_many(thetas, X=Xq)
    assert got.shape == (m, dims[-1], Xq.shape[0])
    for i in range(m):
        np.testing.assert_allclose(got[i], o.forward(spec, thetas[i], Xq, np.float64), rtol=2e-5, atol=2e-5)
    ch.set_validation(Xq, Y[: Xq.shape[0]])
    np.testing.assert_array_equal(ch.forward_many(thetas, which=1), got)       # staged rows: same kernel, same numbers
    np.testing.assert_allclose(ch.forward_many(thetas[:2], which=0)[1], o.forward(spec, thetas[1], X, np.float64), rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(ch.get_state(), theta)
    ch.close()


@pytest.mark.parametrize("dims,prior,judge", [
    ([5, 50, 50, 50, 1], o.PRIOR_CAUCHY, None),                         # the chain's own priors
    ([5, 50, 50, 50, 1], o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN),              # reweight: judged under the other family
    ([20, 100, 100, 2], o.PRIOR_GAUSSIAN, None),
    ([3, 7, 5, 2], o.PRIOR_GAUSSIAN, o.PRIOR_CAUCHY),
])
def test_hyper_probs_many(native, dims, prior, judge):
    """tbnn_hyper_probs_many (predictor.trainProbs / reweight, predictor.py:188-206): for m saved (theta, eta) pairs the sum over
    the dense layers of calculateHyperProbs, one launch -- against the fp64 oracle's layer_hyper_log_prob summed per network,
    <= 4e-6 relative"""
    spec, X, Y, theta, eta = o.synth_problem(dims, 64, o.ACT_RELU, prior, o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, kernel=native.KERNEL_AUTO)
    rng = np.random.default_rng(17)
    m = 23
    thetas = np.zeros((m, theta.size + 3), dtype=np.float32)             # a stride wider than P
    thetas[:, :theta.size] = theta[None, :] * (1.0 + 0.2 * rng.standard_normal((m, theta.size)))
    etas = (eta[None, :] * (1.0 + 0.05 * rng.standard_normal((m, eta.size))) + 0.01 * rng.standard_normal((m, eta.size))).astype(np.float32)
    got = ch.hyper_probs_many(np.ascontiguousarray(thetas[:, :theta.size]), etas, priors=None if judge is None else [judge] * len(layers))
    jspec = spec if judge is None else o.make_spec(dims, o.ACT_RELU, judge, o.LIK_GAUSSIAN, o.ACT_NONE)
    for i in range(m):
        parts = o.unflatten(jspec, thetas[i, :theta.size].astype(np.float64))
        want = sum(float(o.layer_hyper_log_prob(l, etas[i, 4 * k:4 * k + 4].astype(np.float64), W, b, np.float64))
                   for k, (l, (W, b)) in enumerate(zip(jspec.layers, parts)))
        assert abs(got[i] - want) <= 4e-6 * abs(want) + 1e-6, (i, got[i], want)
    ch.close()


def test_reweight_on_device(native, tmp_path):
    """predictor.reweight with the built-in dense layers: trainProbs and the new architecture's sums come from
    tbnn_hyper_probs_many; the normalised weights against a direct evaluation with the oracle"""
    from tensorbnn_amd.predictor import predictor
    rng = np.random.default_rng(11)
    shapes = [(4, 3), (4, 1), (2, 4), (2, 1)]
    names = ["dense", "relu", "dense"]
    w = o.SampleWriter(str(tmp_path / "run"), shapes, names, 9, 1, 1, 2)
    for it in range(1, 7):
        sts = [(0.5 * rng.standard_normal(sh)).astype(np.float32) for sh in shapes]
        hyp = [np.float32(v + 0.02 * rng.standard_normal(1)) for v in ([0.02, 1.0, -0.02, 0.99] * 2 + [0.3])]
        w.after_epoch(it, sts, hyp)
    w.close()
    p = predictor(str(tmp_path / "run") + "/")
    (tmp_path / "arch2.txt").write_text("denseGaussian\nrelu\ndenseGaussian\n")
    wts = p.reweight(str(tmp_path / "arch2.txt"), n=1, likelihood=None)
    assert p._chain is not None
    assert wts.shape == (p.numNetworks,) and abs(wts.sum() - 1) < 1e-5 and np.all(wts >= 0)
    sc = o.make_spec([3, 4, 2], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, o.ACT_NONE)
    sg = o.make_spec([3, 4, 2], o.ACT_RELU, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN, o.ACT_NONE)
    lw = []
    for k in range(p.numNetworks):
        t = [m_[k].astype(np.float64) for m_ in p.matrices]
        h = np.asarray(p.hypers[k], dtype=np.float64).reshape(-1)
        old = sum(float(o.layer_hyper_log_prob(sc.layers[j], h[4 * j:4 * j + 4], t[2 * j], t[2 * j + 1], np.float64)) for j in range(2))
        new = sum(float(o.layer_hyper_log_prob(sg.layers[j], h[4 * j:4 * j + 4], t[2 * j], t[2 * j + 1], np.float64)) for j in range(2))
        lw.append(new - old)
    ref = np.exp(np.array(lw) - max(lw)); ref /= ref.sum()
    np.testing.assert_allclose(wts, ref, rtol=2e-4, atol=1e-7)
    assert [l.name for l in p.layers] == names
