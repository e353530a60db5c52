"""GPU parity of the fused kernels for layer-0 fan-in above 32 (round 4): the tall-fan-in family (kernels_tall.hpp: the fan-in
split over the four waves of a workgroup, one 16-row tile per workgroup -- the reference's MNIST example 784 -> 20 -> 20 -> 1,
docs/ClassificationExample.md:103-173) and the mid-width family with its fan-in limit lifted (100 -> 50 -> 50 -> 1).

Same test set as the mid-width family: value / gradient per tensor / statistic against the fp64 oracle and the thread-per-row
kernel, ragged row counts, several row tiles per workgroup (TBNN_FAST_GRID), whole transitions with trace, forward and
ensemble forward, determinism.  Tolerances: tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

import tbnn_oracle as o
from test_gpu_parity import LOGP_RTOL, check_logp_grad, make_chain

pytestmark = pytest.mark.gpu

TALL = {
    "mnist": dict(dims=[784, 20, 20, 1], n=1003, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI),
    "tall_t1": dict(dims=[70, 24, 40, 1], n=517, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN),
    "tall_t2": dict(dims=[128, 16, 2], n=333, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    "tall_t3": dict(dims=[200, 33, 18, 50, 2], n=900, act=o.ACT_SIGMOID, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI),
}


def problem(case):
    c = TALL[case]
    spec, X, Y, theta, eta = o.synth_problem(c["dims"], c["n"], c["act"], c["prior"], c["lik"])
    if c["dims"][0] >= 100:          # keep the pre-activations of a long fan-in O(1): pixel-like rows in [0, 1) / sqrt(fan-in)
        X = (np.abs(X) / np.sqrt(c["dims"][0])).astype(np.float32)
        if c["lik"] == o.LIK_BERNOULLI:
            Y = (np.random.default_rng(9).random(Y.shape) < 0.5).astype(np.float32)
    return spec, X, Y, theta, eta


@pytest.mark.parametrize("case", list(TALL))
def test_logp_grad_tall(native, case):
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_FAST)
    assert ch.kernel_name.startswith("tall<"), ch.kernel_name
    ch.close()
    lp, g = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    lp_g, g_g = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_GENERIC)
    assert abs(lp - lp_g) <= 2e-6 * abs(lp_g) + 1e-4
    assert np.abs(g - g_g).max() <= 5e-5 * np.abs(g_g).max()


@pytest.mark.parametrize("n", [1, 17, 5000 + 3, 12000 + 7, 20000 + 1, 50000 + 3])
def test_logp_grad_tall_group_sizes_aligned_rows(native, n):
    """the 16-byte-aligned row path (784 columns) at row counts that take groups of 1 (<= 17 rows), 2 (5,003), 3 (12,007; 20,001) and 4
    (50,003) row tiles (tall_group_tiles), each with a ragged last tile and a last group that is not full"""
    spec, X, Y, theta, eta = o.synth_problem([784, 20, 20, 1], n, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI)
    X = (np.abs(X) / 28.0).astype(np.float32)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("case", ["mnist", "tall_t1", "tall_t3"])
def test_tall_every_group_size_on_the_same_rows(native, monkeypatch, case):
    """TBNN_TALL_G forces the instantiation: groups of 1, 2, 3 and 4 row tiles on the SAME rows (a tile count no group size divides) give the
    same value and gradient up to fp32 summation order, and each matches the fp64 oracle"""
    spec, X, Y, theta, eta = problem(case)
    res = []
    for g in (1, 2, 3, 4):
        monkeypatch.setenv("TBNN_TALL_G", str(g))
        res.append(check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST))
    monkeypatch.delenv("TBNN_TALL_G")
    lp0, g0 = res[0]
    for lp, g in res[1:]:
        assert abs(lp - lp0) <= 1e-6 * abs(lp0) + 1e-5
        assert np.abs(g - g0).max() <= 2e-5 * np.abs(g0).max()


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 129, 4200 + 3])
def test_logp_grad_tall_ragged_rows(native, n):
    spec, X, Y, theta, eta = o.synth_problem([70, 24, 40, 1], n, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("case,grid", [("mnist", 3), ("mnist", 64), ("tall_t1", 2), ("tall_t2", 5), ("tall_t3", 7)])
def test_tall_several_tiles_per_workgroup(native, monkeypatch, case, grid):
    """a small grid (TBNN_FAST_GRID): every workgroup walks several groups of row tiles (accumulators carried over the groups, the
    exchange buffer and the delta_0 blocks re-used behind the barriers) -- same value and gradient as the full grid, and as the fp64 oracle"""
    spec, X, Y, theta, eta = problem(case)
    lp0, g0 = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    monkeypatch.setenv("TBNN_FAST_GRID", str(grid))
    lp, g = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    assert abs(lp - lp0) <= 1e-6 * abs(lp0) + 1e-5
    assert np.abs(g - g0).max() <= 2e-5 * np.abs(g0).max()


@pytest.mark.parametrize("case", list(TALL))
def test_transition_tall(native, monkeypatch, case):
    spec, X, Y, theta, eta = problem(case)
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    monkeypatch.setenv("TBNN_FAST_GRID", "5")
    ch = make_chain(native, spec, native.KERNEL_FAST)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    monkeypatch.delenv("TBNN_FAST_GRID")
    eps = 2e-5
    out = ch.hmc_step(eps, 4, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, theta, eta, X, Y, eps, 4, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    # lar is a difference of two fp32-evaluated log-probs: the band of tests/test_gpu_fullsize.py with its
    # resolution term -- un-normalised rows over a 70..784-wide fan-in put |logp| at 1e5..1e6
    lar_tol = 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 1e-6 * abs(ref.trace_logp[0])
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= lar_tol, (out["log_accept_ratio"], ref.log_accept_ratio, ref.trace_logp[0])
    if abs(ref.log_accept_ratio - float(np.log(0.5))) > 2 * lar_tol:
        assert bool(out["accepted"]) == ref.accepted
    np.testing.assert_allclose(ch.get_state(), ref.theta_proposed if out["accepted"] else theta, rtol=0, atol=2e-6 * max(1.0, np.abs(ref.theta).max()))
    # free-running transitions afterwards: finite, and the chain moves
    outs = ch.hmc_run(eps, 5, 4)
    assert all(np.isfinite(o_["log_accept_ratio"]) for o_ in outs)
    ch.close()


@pytest.mark.parametrize("case", list(TALL))
def test_forward_tall(native, case):
    """network.predict / the ensemble forward on the forward-only instantiation (grid.y = network)"""
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_FAST)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    ref = o.forward(spec, theta, X, np.float64)
    f = ch.predict(0)
    np.testing.assert_allclose(f, ref, rtol=2e-5, atol=2e-5)
    rng = np.random.default_rng(5)
    thetas = (theta[None, :] + 0.05 * rng.standard_normal((5, theta.size))).astype(np.float32)
    fm = ch.forward_many(thetas, X=X[:77])
    for i in range(5):
        np.testing.assert_allclose(fm[i], o.forward(spec, thetas[i], X[:77], np.float64), rtol=2e-5, atol=2e-5)
    ch.close()


def test_tall_determinism(native):
    spec, X, Y, theta, eta = problem("mnist")
    res = []
    for _ in range(2):
        ch = make_chain(native, spec, native.KERNEL_FAST)
        ch.set_data(X, Y)
        res.append(ch.logp_grad(theta, eta))
        ch.close()
    assert res[0][0] == res[1][0]
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_mnist_shape_full_rows(native):
    """784 -> 20 -> 20 -> 1 at the tutorial's scale (12,000 rows): value, statistic and every gradient tensor against the C restatement"""
    import c_oracle
    spec, X, Y, theta, eta = o.synth_problem([784, 20, 20, 1], 12000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI)
    X = (np.abs(X) / 28.0).astype(np.float32)
    ch = make_chain(native, spec, native.KERNEL_AUTO)
    assert ch.kernel_name.startswith("tall<"), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    co = c_oracle.COracle(spec, X, Y)
    lp_c, g_c = co.logp_grad(theta, eta)[:2]
    assert abs(lp - lp_c) <= LOGP_RTOL * abs(lp_c)
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g_c[a:b]).max() <= 1e-4 * max(np.abs(g_c[a:b]).max(), 1e-3)
    ch.close()


# ---- the mid-width family with layer-0 fan-in above 32 (kernels_mid.hpp, MID_MAX_FANIN)
@pytest.mark.parametrize("dims,n,act,lik", [
    ([100, 50, 50, 1], 1003, o.ACT_RELU, o.LIK_GAUSSIAN),
    ([40, 24, 40, 1], 517, o.ACT_TANH, o.LIK_GAUSSIAN),
    ([64, 32, 48, 2], 700, o.ACT_SIGMOID, o.LIK_BERNOULLI),
])
def test_mid_long_fan_in(native, dims, n, act, lik):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, lik)
    ch = make_chain(native, spec, native.KERNEL_AUTO, jit=True)
    assert "mid<" in ch.kernel_name, ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3)
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    ch.set_state(theta); ch.set_hypers(eta)
    out = ch.hmc_step(2e-5, 4, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, theta, eta, X, Y, 2e-5, 4, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    lar_tol = 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 1e-6 * abs(ref.trace_logp[0])       # (resolution term: see test_transition_tall)
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= lar_tol, (out["log_accept_ratio"], ref.log_accept_ratio, ref.trace_logp[0])
    ch.close()


def _free_running(native, dims, n, lik, X_scale, eps, epochs, L, family, jit=None, burn=0):
    """a free-running chain on the HIP kernel against oracle/c with the same draws, each carrying its OWN state forward"""
    import c_oracle
    from test_gpu_fullsize import away_from, lar_tol
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    if X_scale:
        X = (np.abs(X) * X_scale).astype(np.float32)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, jit=jit)
    assert family in ch.kernel_name, ch.kernel_name
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    if burn:                                   # away from the initial state (|logp| ~ 1e7 there: lar is not a meaningful quantity to compare)
        ch.hmc_run(eps, 20, burn)
        theta = ch.get_state()
    co = c_oracle.COracle(spec, X, Y)
    rng = np.random.default_rng(2025)
    th_c = theta.copy()
    acc_g, acc_c, dlar, agree, dist = [], [], [], 0, 0.0
    for ep in range(epochs):
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        q_c, lar_c, lp0_c, _ = co.hmc_propose(th_c, eta, eps, L, p0)
        lu = away_from(rng, lar_c)
        if lu < lar_c:
            th_c = q_c
        out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
        agree += int(bool(out["accepted"]) == (lu < lar_c))
        # lar is a function of the state the epoch starts from: the two values are held against the fp32 band while the two chains
        # are at the same state to fp32 resolution.  Two fp32 evaluations with different summation orders drift apart over free-running
        # epochs (a relu unit flipping on one side is enough: 784 -> 20 -> 20 -> 1 goes from 2e-7 to 5e-4 in 30 epochs, with the fused
        # kernel's fringe units on the 4x4x1 MFMA as without); what is compared THROUGHOUT is every decision and the accept ratio
        if np.isfinite(lar_c) and lar_c > -50.0:                       # (a diverged trajectory: both arms reject, the value is not compared)
            if dist <= 2e-5:
                dlar.append(abs(out["log_accept_ratio"] - lar_c) / lar_tol(lar_c, lp0_c))
        else:
            assert not out["accepted"]
        acc_g.append(out["accept_prob"]); acc_c.append(min(1.0, float(np.exp(min(lar_c, 0.0)))))
        dist = float(np.abs(ch.get_state() - th_c).max() / np.abs(th_c).max())
    mg, mc = float(np.mean(acc_g)), float(np.mean(acc_c))
    print(f"free-running {dims} on {ch.kernel_name}: accept ratio HIP {mg:.4f} oracle/c {mc:.4f}; decisions {agree}/{epochs}; "
          f"max |dlar| / lar_tol {max(dlar):.3f} over the {len(dlar)} epochs entered at the same state; final state distance {dist:.1e}")
    ch.close()
    assert abs(mg - mc) <= 0.02, (mg, mc)
    assert agree == epochs and len(dlar) >= min(10, epochs) and max(dlar) <= 1.0, (agree, len(dlar), max(dlar) if dlar else None)
    return mg, mc


def test_free_running_mnist_shape_on_the_tall_kernel(native):
    """north_star's "accept-ratio parity +-0.02" on the round's new family: 784 -> 20 -> 20 -> 1, 12,000 pixel-like rows, 30 free-running
    epochs of L = 10 against the C restatement (same p0 and log u; every decision equal)"""
    _free_running(native, [784, 20, 20, 1], 12000, o.LIK_BERNOULLI, 1.0 / 28.0, 5e-3, 30, 10, "tall<")


def test_free_running_long_fan_in_on_the_mid_kernel(native):
    """the mid-width family beyond fan-in 32: 100 -> 50 -> 50 -> 1 at BASELINE configs[1]'s row count (1e5), 12 free-running epochs
    after 60 burn-in epochs on the HIP chain (both arms start from the burned state)"""
    _free_running(native, [100, 50, 50, 1], 100_000, o.LIK_GAUSSIAN, None, 2e-5, 12, 5, "mid<", jit=True, burn=60)
