"""GPU: the wide family (k_chain_wide + k_dw_wide) behind a first layer of 33 .. 128 inputs (late round 6).  jit.families kept the family to a fan-in of 32 --
round 1's choice for x in registers, W_0's operand granules in LDS and dW_0's accumulator tiles in AccVGPRs -- and a network like 50 -> 100 -> 100 -> 1 or
100 -> 100 -> 100 -> 1 (too many dW tiles for the mid-width kernel, hidden layers too wide for the tall one) ran on the layered family.  The kernels build and
hold for fan-in up to 128 wherever W_0 fits the LDS next to the weight ring (jit.wide_fits): 50 -> 100 -> 100 -> 1 at 1e5 rows 152 against 199 us per
leapfrog step, 40 -> 200 -> 200 -> 1 330 against 480, 128 -> 100 -> 100 -> 10 207 against 261.  No kernel source changed.
Against the fp64 oracle through the C ABI: value, gradient per tensor, forward, every launch three times bit for bit; injected transitions with both
decisions, a hyper transition, free-running epochs on the device's draws (oracle set back on the device's state each epoch)."""
import numpy as np
import pytest

import tbnn_oracle as o
from test_gpu_freerun import Tally, draws, layers_of, SEED
from test_gpu_layered import scaled_problem

pytestmark = pytest.mark.gpu

R, T, S, E = o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU
CASES = {
    # dims, rows, hidden activations, prior, likelihood
    "fanin_50": ([50, 100, 100, 1], 2000, [R, R], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "fanin_64_bern": ([64, 128, 128, 2], 3001, [T, T], o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
    "fanin_40_streamed": ([40, 200, 200, 1], 1500, [E, E], o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),           # weights streamed (not LDS-resident)
    # (tanh: with relu this seeded problem has ONE (row, unit) pre-activation of layer 0 within fp32 rounding of 0 -- unit 43's W_0 row and bias then
    # differ from fp64 by that row's contribution, 1.3e-3 of the tensor's norm, on this kernel and not on the fp32 NumPy oracle: the summation order
    # decides the sign; tools/experiments/wfdbg.py)
    "fanin_100": ([100, 100, 100, 1], 2000 + 9, [T, T], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "fanin_100_relu": ([100, 100, 100, 1], 1777, [R, R], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "fanin_128_ten_outputs": ([128, 100, 100, 10], 3000, [S, S], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "fanin_48_mixed_three_middle": ([48, 90, 130, 70, 3], 777, [T, R, R], o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "fanin_33_few_rows": ([33, 80, 96, 1], 19, [R, T], o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
}


def problem(name):
    dims, n, acts, prior, lik = CASES[name]
    return scaled_problem(dims, n, acts, prior, lik)


def chain(native, monkeypatch, name, spec, **kw):
    monkeypatch.setenv("TBNN_JIT_SKIP", "fast3,fast,mid,tall")
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True, **kw)
    assert ch.kernel_name.startswith("jit-wide"), ch.kernel_name
    return ch


@pytest.mark.parametrize("name", list(CASES))
def test_value_gradient_forward(native, monkeypatch, name):
    spec, X, Y, theta, eta = problem(name)
    assert 32 < spec.layers[0].in_dim <= 128
    ch = chain(native, monkeypatch, name, spec)
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    for _ in range(2):
        lp2, g2, _s = ch.logp_grad(theta, eta)
        assert lp2 == lp and np.array_equal(g, g2)
    m = min(500, X.shape[0])
    f = ch.forward(X[:m], theta)
    assert np.array_equal(f, ch.forward(X[:m], theta))
    ch.close()
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    assert abs(lp - lp64) <= 4e-6 * max(abs(lp64), 1.0), (lp, lp64)
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (name, a, b)
    f64 = o.forward(spec, theta, X[:m], np.float64)
    assert np.abs(f - f64).max() <= 1e-4


@pytest.mark.parametrize("name", ["fanin_50", "fanin_64_bern", "fanin_40_streamed", "fanin_128_ten_outputs", "fanin_48_mixed_three_middle"])
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
        for ep in range(10):
            rec = ch.hmc_run(1e-4, 5, 1)[0]
            p0e, lu = draws(spec.n_params, 2, ep)
            ref = o.weight_step(spec, th, eta, X, Y, 1e-4, 5, p0e, lu, np.float64)
            took = t.add(rec, ref.log_accept_ratio, lu, ref.logp_old)
            want = ref.theta_proposed.astype(np.float64) if took else th
            got = ch.get_state().astype(np.float64)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), ep
            th = got
    ch.close()
    t.check(f"wide family, fan-in above 32 [{name}]")
