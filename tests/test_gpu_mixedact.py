"""GPU: hidden layers with DIFFERENT activations on the fused kernels (round 6).  network.add takes any sequence of dense and activation layers
(tensorBNN/network.py:173-191; activationFunctions.py:27-75): Relu behind one layer and Tanh behind the next is an ordinary network there.  Until
round 6 such a stack ran on the layered family only (jit.shape_of: "one activation for all hidden layers"); the fused kernels' Shape now carries a
per-layer code (csrc/kernels_fast.hpp: Shape::act) and every family instantiates it.  Against the fp64 oracle through the C ABI, on each of the
five fused kernels (narrow fast3 / fast, mid, tall, wide): value, gradient per tensor, forward, every launch repeated bit for bit; an injected
transition with both decisions + a hyper transition; 12 free-running epochs on the device's draws with the oracle set back on the device's state
each epoch (test_gpu_freerun); a chain group against its solo chains."""
import numpy as np
import pytest

import tbnn_oracle as o
from test_gpu_freerun import Tally, draws, layers_of, SEED
from test_gpu_layered import scaled_problem

pytestmark = pytest.mark.gpu

R, T, S, E, X_, N = o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU, o.ACT_EXP, o.ACT_NONE
CASES = {
    # dims, rows, hidden activations, prior, likelihood, family, kernel-name prefix
    "fast3_relu_tanh": ([5, 20, 24, 1], 3001, [R, T], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "narrow", "jit-fast3<relu+tanh,"),
    "fast3_three": ([5, 50, 50, 50, 1], 4000, [T, R, E], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "narrow", "jit-fast3<tanh+relu+elu,"),   # configs[1]'s dims
    "fast_none_sigmoid": ([4, 17, 9, 3], 555, [N, S], o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN, "fast", "jit-fast<none+sigmoid,"),                 # MFMA last layer
    "mid_tanh_relu": ([20, 64, 64, 2], 5000, [T, R], o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "mid", "jit-mid<tanh+relu,"),
    "mid_relu_sigmoid_elu": ([12, 40, 33, 48, 5], 1234, [R, S, E], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "mid", "jit-mid<relu+sigmoid+elu,"),    # 5 outputs
    "tall_elu_sigmoid": ([100, 50, 50, 1], 3005, [E, S], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "tall", "jit-tall<elu+sigmoid,"),
    "tall_relu_tanh_10": ([300, 20, 20, 10], 1205, [R, T], o.PRIOR_CAUCHY, o.LIK_BERNOULLI, "tall", "jit-tall<relu+tanh,"),
    "wide_relu_sigmoid": ([10, 200, 120, 1], 4000 + 7, [R, S], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, "wide", "jit-wide<relu+sigmoid,"),
    "wide_tanh_relu_relu": ([8, 90, 130, 70, 2], 2000, [T, R, R], o.PRIOR_GAUSSIAN, o.LIK_BERNOULLI, "wide", "jit-wide<tanh+relu+relu,"),
}
SKIP = {"narrow": "", "fast": "mid,tall,wide", "mid": "fast3,fast,tall,wide", "tall": "fast3,fast,mid,wide", "wide": "fast3,fast,mid,tall"}


def named(kernel_name, prefix):
    """prefix = family<activations,  -- the wide family puts "(resident)" between the two for small shapes"""
    fam, acts = prefix.split("<")
    return kernel_name.startswith(fam) and ("<" + acts) in kernel_name


def problem(name):
    dims, n, acts, prior, lik = CASES[name][:5]
    return scaled_problem(dims, n, acts, prior, lik)


def chain(native, monkeypatch, name, spec, **kw):
    fam, prefix = CASES[name][5:]
    monkeypatch.setenv("TBNN_JIT_SKIP", SKIP[fam])
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True, **kw)
    assert named(ch.kernel_name, prefix), ch.kernel_name
    return ch


@pytest.mark.parametrize("name", list(CASES))
def test_value_gradient_forward(native, monkeypatch, name):
    spec, X, Y, theta, eta = problem(name)
    assert len({l.act for l in spec.layers[:-1]}) > 1
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
    assert np.abs(f - f64).max() <= 1e-4


def test_the_layer_order_matters(native, monkeypatch):
    """relu+tanh and tanh+relu are two kernels with two answers (a code that lost the layer index would pass every single-order test)"""
    dims, n = [5, 20, 24, 1], 1000
    out = {}
    for acts in ([R, T], [T, R]):
        spec, X, Y, theta, eta = scaled_problem(dims, n, acts, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
        ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, jit=True)
        assert ch.kernel_name.startswith("jit-fast3<" + ("relu+tanh," if acts[0] == R else "tanh+relu,")), ch.kernel_name
        ch.set_data(X, Y)
        lp, g, _ = ch.logp_grad(theta, eta)
        ch.close()
        lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
        assert abs(lp - lp64) <= 4e-6 * max(abs(lp64), 1.0)
        assert np.abs(g - g64).max() <= 1e-4 * np.abs(g64).max()
        out[tuple(acts)] = lp
    assert abs(out[(R, T)] - out[(T, R)]) > 1e-3 * abs(out[(R, T)])


@pytest.mark.parametrize("name", ["fast3_three", "mid_tanh_relu", "tall_relu_tanh_10", "wide_relu_sigmoid"])
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
    ch.set_state(theta); ch.set_hypers(eta); ch.set_epoch(0)
    t, th = Tally(), theta.astype(np.float64)
    with np.errstate(all="ignore"):
        for ep in range(12):
            rec = ch.hmc_run(2e-4, 5, 1)[0]
            p0e, lu = draws(spec.n_params, 2, ep)
            ref = o.weight_step(spec, th, eta, X, Y, 2e-4, 5, p0e, lu, np.float64)
            took = t.add(rec, ref.log_accept_ratio, lu, ref.logp_old)
            want = ref.theta_proposed.astype(np.float64) if took else th
            got = ch.get_state().astype(np.float64)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), ep
            th = got
    ch.close()
    t.check(f"mixed activations [{name}]")


def test_chain_group_equals_solo_chains(native, monkeypatch):
    name = "mid_tanh_relu"
    spec, X, Y, theta, eta = problem(name)
    fam, prefix = CASES[name][5:]
    monkeypatch.setenv("TBNN_JIT_SKIP", SKIP[fam])
    C, eps, L, EP = 3, 2e-4, 4, 5
    rng = np.random.default_rng(8)
    thetas = (theta[None, :] * (1.0 + 0.03 * rng.standard_normal((C, theta.size)))).astype(np.float32)
    etas = np.tile(eta, (C, 1)).astype(np.float32)
    grp = native.ChainGroup(layers_of(spec), C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=SEED, chain_id=4, jit=True)
    assert named(grp.kernel_name, prefix), grp.kernel_name
    grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(etas)
    recs = grp.hmc_run(eps, L, EP)
    states = grp.get_state()
    grp.close()
    for c in range(C):
        ch = chain(native, monkeypatch, name, spec, seed=SEED, chain_id=4 + c)
        ch.set_data(X, Y); ch.set_state(thetas[c]); ch.set_hypers(etas[c])
        solo = ch.hmc_run(eps, L, EP)
        assert [r["log_accept_ratio"] for r in solo] == [r["log_accept_ratio"] for r in recs[c]]
        assert np.array_equal(ch.get_state(), states[c])
        ch.close()
