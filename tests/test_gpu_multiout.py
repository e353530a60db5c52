"""GPU: networks with MORE THAN TWO outputs on a fused kernel (round 6; mid-width, tall and -- late in the round -- wide family).  network.add takes any stack of layers (tensorBNN/network.py:173-191),
GaussianLikelihood / BernoulliLikelihood sum over [d_out, n] (likelihood.py:88-94, 226-236); until round 6 the mid-width and tall families ran their last
layer on the VALU (<= 2 outputs) and every wider output went to the layered family.  3 .. 16 outputs: the last layer is one more MFMA layer of
k_fwd_bwd_mid / k_fwd_bwd_tall (one output tile, the likelihood reads the tile) -- 784 -> 20 -> 20 -> 10 is the reference's MNIST tutorial
(docs/ClassificationExample.md:103-173) with all ten digits.  Against the fp64 oracle through the C ABI: value, gradient per tensor, forward,
an injected transition with both decisions, a hyper transition, 20 free-running epochs on the device's draws (the oracle set back on the device's
state every epoch: test_gpu_freerun), ragged row counts, every launch repeated bit for bit."""
import numpy as np
import pytest

import tbnn_oracle as o
from test_gpu_freerun import Tally, draws, layers_of, SEED

pytestmark = pytest.mark.gpu

CASES = {
    # dims, rows, hidden activation, prior, likelihood, family
    "gauss5": ([20, 64, 64, 5], 4000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "mid"),
    "bern10": ([30, 80, 80, 10], 3001, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "mid"),         # ragged last tile
    "tanh3_two_middle": ([7, 33, 18, 50, 3], 777, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN, "mid"),
    "bern16_full_tile": ([12, 40, 48, 16], 1234, o.ACT_ELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "mid"),
    "few_rows": ([20, 64, 64, 5], 9, o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "mid"),
    "mnist10": ([784, 20, 20, 10], 1205, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "tall"),      # groups of 1 .. 4 tiles, ragged
    "tall_gauss5": ([100, 50, 50, 5], 5000, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN, "tall"),
    "tall_one_hidden3": ([300, 33, 3], 333, o.ACT_ELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "tall"),     # the last layer is the only MFMA layer in LDS
    # the wide family (k_chain_wide + k_dw_wide): the last layer as one more middle layer -- streamed weights, a_LL / delta_LL through HBM
    "wide10": ([10, 200, 200, 10], 3001, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "wide"),     # configs[3]'s widths, ten classes
    "wide5_resident": ([20, 100, 100, 5], 4000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "wide"),   # configs[4]'s widths (image resident in LDS)
    "wide3_three_middle": ([8, 90, 130, 70, 3], 777, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN, "wide"),
    "wide16_full_tile": ([12, 80, 96, 16], 1234, o.ACT_ELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "wide"),
}
SKIP = {"mid": "fast3,fast,tall,wide", "tall": "fast3,fast,mid,wide", "wide": "fast3,fast,mid,tall"}


def problem(name):
    dims, n, act, prior, lik, _fam = CASES[name]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    if dims[0] > 64:
        X = (X / np.sqrt(dims[0] / 16.0)).astype(np.float32)          # keep a long fan-in's pre-activations O(1)
    if lik == o.LIK_BERNOULLI:
        theta = (theta * 0.3).astype(np.float32)          # outputs off saturation: a well-conditioned fp32 problem
    return spec, X, Y, theta, eta


def chain(native, monkeypatch, name, spec, **kw):
    fam = CASES[name][5]
    monkeypatch.setenv("TBNN_JIT_SKIP", SKIP[fam])
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True, **kw)
    assert ch.kernel_name.startswith(f"jit-{fam}"), ch.kernel_name       # (jit-wide(resident)< for images that fit in LDS)
    return ch


@pytest.mark.parametrize("name", list(CASES))
def test_value_gradient_forward(native, monkeypatch, name):
    spec, X, Y, theta, eta = problem(name)
    assert spec.layers[-1].out_dim > 2
    ch = chain(native, monkeypatch, name, spec)
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    for _ in range(2):
        lp2, g2, _s = ch.logp_grad(theta, eta)
        assert lp2 == lp and np.array_equal(g, g2)
    f = ch.forward(X[:500], theta)
    assert np.array_equal(f, ch.forward(X[:500], theta))
    ch.close()
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    assert abs(lp - lp64) <= 4e-6 * max(abs(lp64), 1.0), (lp, lp64)
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (name, a, b)
    f64 = o.forward(spec, theta, X[:500], np.float64)
    assert f.shape == f64.shape == (spec.layers[-1].out_dim, min(500, X.shape[0]))
    assert np.abs(f - f64).max() <= 1e-4


@pytest.mark.parametrize("name", ["gauss5", "bern10", "tanh3_two_middle", "mnist10", "tall_gauss5", "wide10", "wide5_resident", "wide3_three_middle"])
def test_transitions(native, monkeypatch, name):
    spec, X, Y, theta, eta = problem(name)
    rng = np.random.default_rng(4)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    ch = chain(native, monkeypatch, name, spec, seed=SEED, chain_id=2)
    ch.set_data(X, Y)
    lp64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[0]
    for log_u in (-1e30, 1e30):
        ch.set_state(theta); ch.set_hypers(eta)
        out = ch.hmc_step(3e-5, 4, p0=p0, log_u=log_u)
        ref = o.weight_step(spec, theta, eta, X, Y, 3e-5, 4, p0, log_u, np.float64)
        assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 4e-7 * abs(lp64)
        assert bool(out["accepted"]) == ref.accepted
        assert np.abs(ch.get_state() - ref.theta).max() <= 1e-5 * max(1.0, np.abs(ref.theta).max())
    ph = rng.standard_normal(spec.n_hypers).astype(np.float32)
    ch.set_state(theta); ch.set_hypers(eta)
    ch.logp_grad(theta, eta)
    out = ch.hyper_step(1e-4, 9, p0=ph, log_u=-1e30)
    ref = o.hyper_step(spec, eta, theta, X, Y, 1e-4, 9, ph, -1e30, np.float64)
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-3 * abs(ref.log_accept_ratio)
    assert np.allclose(ch.get_hypers(), ref.theta, rtol=1e-4, atol=1e-5)
    # 20 epochs on the device's own draws, the oracle set back on the device's state every epoch
    ch.set_state(theta); ch.set_hypers(eta); ch.set_epoch(0)
    t, th = Tally(), theta.astype(np.float64)
    eps = 1e-3 if name in ("bern10", "mnist10", "wide10") else 2e-4
    with np.errstate(all="ignore"):
        for ep in range(20):
            rec = ch.hmc_run(eps, 5, 1)[0]
            p0e, lu = draws(spec.n_params, 2, ep)
            ref = o.weight_step(spec, th, eta, X, Y, eps, 5, p0e, lu, np.float64)
            took = t.add(rec, ref.log_accept_ratio, lu, ref.logp_old)
            want = ref.theta_proposed.astype(np.float64) if took else th
            got = ch.get_state().astype(np.float64)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), ep
            th = got
    ch.close()
    t.check(f"multi-output [{name}]")


@pytest.mark.parametrize("name", ["gauss5", "mnist10", "wide10"])
def test_chain_group_equals_solo_chains(native, monkeypatch, name):
    """several chains behind one handle (gridDim.y = chain) on the multi-output kernels: chain c is the solo chain chain_id + c, bit for bit"""
    spec, X, Y, theta, eta = problem(name)
    fam = CASES[name][5]
    monkeypatch.setenv("TBNN_JIT_SKIP", SKIP[fam])
    C, eps, L, E = 3, (1e-3 if name in ("mnist10", "wide10") else 2e-4), 4, 5
    rng = np.random.default_rng(8)
    thetas = (theta[None, :] * (1.0 + 0.03 * rng.standard_normal((C, theta.size)))).astype(np.float32)
    etas = np.tile(eta, (C, 1)).astype(np.float32)
    grp = native.ChainGroup(layers_of(spec), C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=SEED, chain_id=4, jit=True)
    assert grp.kernel_name.startswith(f"jit-{fam}")
    grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(etas)
    recs = grp.hmc_run(eps, L, E)
    states = grp.get_state()
    grp.close()
    for c in range(C):
        ch = chain(native, monkeypatch, name, spec, seed=SEED, chain_id=4 + c)
        ch.set_data(X, Y); ch.set_state(thetas[c]); ch.set_hypers(etas[c])
        solo = ch.hmc_run(eps, L, E)
        assert [r["log_accept_ratio"] for r in solo] == [r["log_accept_ratio"] for r in recs[c]]
        assert [r["accepted"] for r in solo] == [r["accepted"] for r in recs[c]]
        assert np.array_equal(ch.get_state(), states[c])
        ch.close()
