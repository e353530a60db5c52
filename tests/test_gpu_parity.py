"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerances (BASELINE.md section 5 / north_star "stated fp32 tolerance"):
  * target log-prob: <= 4e-6 relative (+ 1e-3 absolute for tiny values)
  * gradient: <= 1e-4 relative to the tensor's inf-norm
  * accept decision / log-accept-ratio with injected p0, log u: |dlar| <= 2e-2 + 1e-4*|lar|
The oracle arm is float64 (the exact value of the reference's formula); the
float32 oracle arm (the reference's own arithmetic) must sit within the same
band, which the CPU suite checks (tests/test_oracle.py).
"""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

LOGP_RTOL = 4e-6
GRAD_RTOL = 1e-4


def make_chain(native, spec, kernel=None, **kw):
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd,
                        kernel=native.KERNEL_AUTO if kernel is None else kernel, **kw)


def check_logp_grad(native, spec, X, Y, theta, eta, kernel=None):
    ch = make_chain(native, spec, kernel)
    ch.set_data(X, Y)
    lp, g, stat = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3, (ch.kernel_name, lp, lp64)
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            ref = g64[a:b]
            err = np.abs(g[a:b] - ref).max()
            assert err <= GRAD_RTOL * max(np.abs(ref).max(), 1e-3), (ch.kernel_name, a, b, err, np.abs(ref).max())
    ch.close()
    return lp, g


CASES = {
    # BASELINE configs[0] (plumbing): 1->10->10->1, GaussianLikelihood, 1k rows
    "c1": dict(dims=[1, 10, 10, 1], n=1000, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    # literal Examples/trainRegression.py shape: 1->10->10->10->1 Tanh, GaussianDenseLayer, FixedGaussian, 11 rows
    "trainreg": dict(dims=[1, 10, 10, 10, 1], n=11, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_FIXED_GAUSSIAN),
    # down-scaled configs[1]
    "c2_small": dict(dims=[5, 50, 50, 50, 1], n=1000, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    # ragged: n not a multiple of any tile
    "c2_ragged": dict(dims=[5, 50, 50, 50, 1], n=777, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    # down-scaled configs[4]: Bernoulli + sigmoid
    "c5_small": dict(dims=[20, 100, 100, 2], n=600, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI),
    # down-scaled configs[3]
    "c4_small": dict(dims=[10, 200, 200, 200, 1], n=300, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    # single row, single layer
    "tiny": dict(dims=[3, 1], n=1, act=o.ACT_NONE, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN),
    "sigmoid_hidden": dict(dims=[4, 7, 3], n=130, act=o.ACT_SIGMOID, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
    "elu_hidden": dict(dims=[3, 20, 17, 2], n=333, act=o.ACT_ELU, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN),
    "exp_hidden": dict(dims=[2, 6, 1], n=70, act=o.ACT_EXP, prior=o.PRIOR_CAUCHY, lik=o.LIK_FIXED_GAUSSIAN),
    # wide-path test shapes (kernels_wide.hpp): ragged widths / widths that are multiples of 16
    "wide_t1": dict(dims=[3, 20, 36, 2], n=517, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN),
    "wide_t2": dict(dims=[20, 32, 16, 48, 2], n=1030, act=o.ACT_SIGMOID, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI),
    # mid-width fused kernel's test shapes (kernels_mid.hpp): ragged widths / multiples of 16 / two middle layers
    "mid_t1": dict(dims=[4, 24, 40, 1], n=517, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN),
    "mid_t2": dict(dims=[20, 32, 48, 2], n=1030, act=o.ACT_SIGMOID, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI),
    "mid_t3": dict(dims=[7, 33, 18, 50, 2], n=900, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN),
}


def problem(case):
    c = CASES[case]
    return o.synth_problem(c["dims"], c["n"], c["act"], c["prior"], c["lik"])


@pytest.mark.parametrize("case", list(CASES))
def test_logp_grad_generic(native, case):
    spec, X, Y, theta, eta = problem(case)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_GENERIC)


FAST_CASES = ["c1", "trainreg", "c2_small", "c2_ragged", "sigmoid_hidden"]


@pytest.mark.parametrize("case", FAST_CASES)
def test_logp_grad_fast(native, case):
    """the shape-specialised MFMA kernel (TBNN_KERNEL_FAST must exist for these shapes)"""
    spec, X, Y, theta, eta = problem(case)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


WIDE_CASES = ["wide_t1", "wide_t2", "c5_small", "c4_small"]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_logp_grad_wide(native, case, monkeypatch):
    """the two-kernel wide-layer path (k_chain_wide + k_dw_wide) must cover these shapes (TBNN_MID=0: configs[4]'s shape,
    which the mid-width fused kernel takes by default, stays covered as the wide family's resident-weights case)"""
    monkeypatch.setenv("TBNN_MID", "0")
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_FAST)
    assert ch.kernel_name.startswith("wide<"), ch.kernel_name
    ch.close()
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


MID_CASES = ["mid_t1", "mid_t2", "mid_t3", "c5_small"]


@pytest.mark.parametrize("case", MID_CASES)
def test_logp_grad_mid(native, case):
    """the mid-width fused kernel (k_fwd_bwd_mid: dW of every layer in one wave's AccVGPRs, W^T read from the W image)"""
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_FAST)
    assert ch.kernel_name.startswith("mid<"), ch.kernel_name
    ch.close()
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 129, 5000 + 3])
def test_logp_grad_mid_ragged_rows(native, n):
    spec, X, Y, theta, eta = o.synth_problem([7, 33, 18, 50, 2], n, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("case", ["mid_t1", "mid_t2", "mid_t3", "c5_small"])
def test_transition_mid(native, case):
    """a whole transition through the mid-width kernel: log-prob trace, log accept ratio, decision, new state vs the fp64 oracle;
    several tiles per wave through a small grid (TBNN_FAST_GRID), so that the accumulators carry over tiles"""
    spec, X, Y, theta, eta = problem(case)
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    import os
    os.environ["TBNN_FAST_GRID"] = "3"
    try:
        ch = make_chain(native, spec, native.KERNEL_FAST)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    finally:
        del os.environ["TBNN_FAST_GRID"]
    eps = 2e-5
    out = ch.hmc_step(eps, 4, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, theta, eta, X, Y, eps, 4, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio)
    assert bool(out["accepted"]) == ref.accepted
    th = ch.get_state()
    np.testing.assert_allclose(th, ref.theta, rtol=0, atol=2e-6 * max(1.0, np.abs(ref.theta).max()))
    ch.close()


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 129, 5000 + 3])
def test_logp_grad_wide_ragged_rows(native, n):
    spec, X, Y, theta, eta = o.synth_problem([3, 20, 36, 2], n, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 4096 + 3])
def test_logp_grad_fast_ragged_rows(native, n):
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], n)
    check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)


@pytest.mark.parametrize("dims,act,prior,lik,n,grid", [
    ([5, 50, 50, 50, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 2000, 10),     # 125 tiles = 3 rounds of 40 + 5: one cooperative round, half the workgroups idle
    ([5, 50, 50, 50, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 2000, 3),      # 10 rounds of 12 + 5: two cooperative rounds, the second with one dummy tile
    ([5, 50, 50, 50, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 1991, 10),     # the last cooperative tile is ragged (7 of 16 rows)
    ([5, 50, 50, 50, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 650, 10),      # 41 tiles = 1 round + 1 tile
    ([5, 50, 50, 50, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 2000, 5),      # 6 rounds of 20 + 5: every workgroup has a cooperative tile
    ([1, 10, 10, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, 1000, 5),          # one M tile per layer: wave 0 alone works, the barriers still meet
    ([1, 10, 10, 10, 1], o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN, 700, 4),
])
def test_logp_grad_fast_cooperative_tail(native, monkeypatch, dims, act, prior, lik, n, grid):
    """k_fwd_bwd_fast3's cooperative tail (Coop3: the left-over tiles of the last round on the 4 waves of a workgroup
    together), forced at small row counts by a small grid (TBNN_FAST_GRID); full size: test_fast_vs_generic_full_size,
    tests/test_gpu_fullsize.py.  Value, gradient and statistic against the fp64 oracle and the plain loop (a grid that
    leaves no remainder)."""
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    monkeypatch.setenv("TBNN_FAST_GRID", str(grid))
    lp, g = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    monkeypatch.delenv("TBNN_FAST_GRID")
    lp0, g0 = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    assert abs(lp - lp0) <= 1e-6 * abs(lp0)
    assert np.abs(g - g0).max() <= 2e-5 * np.abs(g0).max()
    # a transition through it: trace of log-probs and the decision
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    monkeypatch.setenv("TBNN_FAST_GRID", str(grid))
    ch = make_chain(native, spec, native.KERNEL_FAST)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    eps = 2e-3 if lik == o.LIK_FIXED_GAUSSIAN else 5e-5
    out = ch.hmc_step(eps, 4, p0=p0, log_u=float(np.log(0.5)), trace=True)
    ref = o.weight_step(spec, theta, eta, X, Y, eps, 4, p0, float(np.log(0.5)), np.float64)
    np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3)
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio)
    ch.close()


@pytest.mark.parametrize("case", ["c1", "c2_small", "c5_small"])
def test_one_launch_transition_end_is_bit_identical(native, monkeypatch, case):
    """The Metropolis decision + host record + commit in ONE k_energy launch (default, P <= 32 k) against the three launches of
    rounds 1-2 (TBNN_MERGE_ENDS=0): the same records and the same state over free-running transitions with accepts and rejects."""
    spec, X, Y, theta, eta = problem(case)
    res = []
    for merged in ("1", "0"):
        monkeypatch.setenv("TBNN_MERGE_ENDS", merged)
        ch = make_chain(native, spec, native.KERNEL_AUTO, seed=21, chain_id=3)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        outs = ch.hmc_run(1e-5, 5, 6) + ch.hmc_run(5e-2, 5, 6) + [ch.hmc_step(1e-5, 3, trace=True)]      # accepts, then rejects
        res.append((ch.get_state(), [(o_["log_accept_ratio"], o_["accepted"], o_["logp_old"], o_["logp_new"], o_["kinetic_new"], o_["sjd"]) for o_ in outs],
                    outs[-1]["trace_logp"]))
        ch.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(np.array(res[0][1], dtype=np.float64), np.array(res[1][1], dtype=np.float64))     # (NaN == NaN: a diverged proposal)
    np.testing.assert_array_equal(res[0][2], res[1][2])
    acc = [r[1] for r in res[0][1]]
    assert 0 < sum(acc) < len(acc), acc


def test_fast_vs_generic_full_size(native):
    """BASELINE configs[1] at full size (n=1e5): MFMA kernel vs generic kernel vs fp64 oracle."""
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 100000)
    lp_f, g_f = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_FAST)
    lp_g, g_g = check_logp_grad(native, spec, X, Y, theta, eta, kernel=native.KERNEL_GENERIC)
    assert abs(lp_f - lp_g) <= 1e-6 * abs(lp_g)
    assert np.abs(g_f - g_g).max() <= 2e-5 * np.abs(g_g).max()


def test_transition_full_size_vs_c_restatement(native):
    """one whole transition at BASELINE configs[1]'s full size (n = 1e5, L = 10, injected momentum and uniform) against the
    C restatement of the reference's fp32 path (oracle/c): per-step log-prob trace, log accept ratio, decision, new state"""
    import c_oracle
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 100000)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood)
    assert ch.kernel_name.startswith("fast3<")
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    rng = np.random.default_rng(77)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    eps, L, lu = 2e-5, 10, float(np.log(0.3))
    out = ch.hmc_step(eps, L, p0=p0, log_u=lu, trace=True)
    co = c_oracle.COracle(spec, X, Y)
    th_c, acc_c, lar_c, lp_old, lp_new = co.hmc_step(theta, eta, eps, L, p0, lu)
    assert abs(out["logp_old"] - lp_old) <= 4e-6 * abs(lp_old)
    assert abs(out["logp_new"] - lp_new) <= 4e-6 * abs(lp_new)
    assert abs(out["log_accept_ratio"] - lar_c) <= 2e-2 + 1e-4 * abs(lar_c)
    assert bool(out["accepted"]) == acc_c
    if acc_c:
        np.testing.assert_allclose(ch.get_state(), th_c, rtol=0, atol=2e-6 * np.abs(th_c).max() + 1e-7)
    ch.close()


@pytest.mark.parametrize("case", list(CASES))
def test_forward(native, case):
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_GENERIC)
    f = ch.forward(X, theta)
    ref = o.forward(spec, theta, X, np.float64)
    assert f.shape == ref.shape
    np.testing.assert_allclose(f, ref, rtol=2e-5, atol=2e-5)
    ch.close()


@pytest.mark.parametrize("kern", ["generic", "auto"])
@pytest.mark.parametrize("case", ["c1", "trainreg", "c2_small", "c5_small"])
def test_hmc_step_injected(native, case, kern):
    """5-step leapfrog trajectory + accept decision with injected p0, log u."""
    spec, X, Y, theta, eta = problem(case)
    kernel = native.KERNEL_GENERIC if kern == "generic" else native.KERNEL_AUTO
    rng = np.random.default_rng(7)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    eps = {"c1": 2e-4, "trainreg": 2e-3, "c2_small": 5e-5, "c5_small": 2e-4}[case]
    L = 5
    for log_u in (np.log(0.5), -1e30, 1e30):
        ch = make_chain(native, spec, kernel)
        ch.set_data(X, Y)
        ch.set_state(theta)
        ch.set_hypers(eta)
        out = ch.hmc_step(eps, L, p0=p0, log_u=log_u, trace=True)
        ref = o.weight_step(spec, theta, eta, X, Y, eps, L, p0, log_u, np.float32, energy_dtype=np.float64)
        ref64 = o.weight_step(spec, theta, eta, X, Y, eps, L, p0, log_u, np.float64)
        tr = np.asarray(out["trace_logp"])
        np.testing.assert_allclose(tr, np.asarray(ref64.trace_logp), rtol=LOGP_RTOL, atol=2e-3)
        assert abs(out["log_accept_ratio"] - ref64.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref64.log_accept_ratio)
        if abs(ref64.log_accept_ratio - log_u) > 0.1:
            assert bool(out["accepted"]) == ref64.accepted
        new = ch.get_state()
        expect = ref64.theta_proposed if out["accepted"] else theta
        np.testing.assert_allclose(new, expect, rtol=0, atol=2e-5 * max(1.0, np.abs(expect).max()))
        if out["accepted"]:
            assert abs(out["sjd"] - ref64.sjd) <= 1e-3 * ref64.sjd + 1e-12
        else:
            assert out["sjd"] == 0.0
        assert abs(out["accept_prob"] - ref64.accept_prob) <= 2e-2
        ch.close()


def test_chain_rng_matches_oracle(native):
    spec, X, Y, theta, eta = problem("c1")
    ch = make_chain(native, spec, native.KERNEL_GENERIC, seed=50, chain_id=3)
    z, lu = ch.debug_draw(epoch=5, purpose=0, n=141)
    ref = o.philox_normals(141, 50, 3, 5, 0)
    np.testing.assert_allclose(z, ref, rtol=0, atol=2e-6)
    assert abs(lu - o.philox_log_uniform(50, 3, 5, 1)) < 1e-6
    ch.close()


def test_hmc_run_matches_steps(native):
    """tbnn_hmc_run (no host round trip) == the same epochs issued one by one."""
    spec, X, Y, theta, eta = problem("c1")
    res = []
    for mode in ("run", "steps"):
        ch = make_chain(native, spec, native.KERNEL_GENERIC, seed=50, chain_id=1)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        if mode == "run":
            outs = ch.hmc_run(2e-4, 7, 6)
        else:
            outs = [ch.hmc_step(2e-4, 7) for _ in range(6)]
        res.append((ch.get_state(), [o_["log_accept_ratio"] for o_ in outs], [o_["accepted"] for o_ in outs]))
        ch.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert any(res[0][2])


@pytest.mark.parametrize("case", ["c1", "trainreg", "c2_small", "c5_small"])
def test_hyper_logp_grad(native, case):
    spec, X, Y, theta, eta = problem(case)
    ch = make_chain(native, spec, native.KERNEL_GENERIC)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    rng = np.random.default_rng(3)
    eta2 = (eta + 0.05 * rng.standard_normal(eta.size)).astype(np.float32)
    lp, g = ch.hyper_logp_grad(eta2)
    lp64, g64 = o.hyper_log_prob_and_grad(spec, eta2, theta, X, Y, np.float64)
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
    np.testing.assert_allclose(g, g64, rtol=2e-4, atol=1e-3 + 2e-6 * np.abs(g64).max())
    ch.close()


@pytest.mark.parametrize("case", ["c1", "trainreg", "c5_small", "tiny"])      # tiny: H = 5 > P = 4 (injected momentum buffer)
def test_hyper_step_injected(native, case):
    spec, X, Y, theta, eta = problem(case)
    rng = np.random.default_rng(11)
    p0 = rng.standard_normal(spec.n_hypers).astype(np.float32)
    for log_u in (-1e30, 1e30):
        ch = make_chain(native, spec, native.KERNEL_GENERIC)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        out = ch.hyper_step(1e-4, 20, p0=p0, log_u=log_u)
        ref = o.hyper_step(spec, eta, theta, X, Y, 1e-4, 20, p0, log_u, np.float64)
        assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-3 * abs(ref.log_accept_ratio)
        assert bool(out["accepted"]) == ref.accepted
        np.testing.assert_allclose(ch.get_hypers(), ref.theta, rtol=1e-4, atol=1e-5)
        ch.close()
