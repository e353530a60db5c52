"""Pins the CPU oracle (oracle/) -- runs without a GPU.

The reference supplies no golden vectors and cannot be imported (no TensorFlow),
so the oracle is pinned by: scipy closed forms for the density helpers (with the
reference's quirks Q1/Q2 asserted as deltas), torch.autograd (fp64) for every
hand-coded gradient, HMC invariants, the committed fixtures (tests/golden) and
agreement between the NumPy and the C restatement.
"""
import math
import os

import numpy as np
import pytest
import scipy.stats as st

import tbnn_oracle as o

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- density KATs
def test_cauchy_log_prob_vs_scipy_with_sign_quirk():
    """BNN_functions.py:51-55 returns +log(1+z^2) - log(pi*gamma); scipy's cauchy.logpdf is
    -log(1+z^2) - log(pi*gamma).  Their sum is -2 log(pi*gamma) (Q1)."""
    x = np.linspace(-3, 3, 13)
    gamma, x0 = 0.5, 0.1
    ref = o.cauchy_log_prob(gamma, x0, x, np.float64)
    sp = st.cauchy.logpdf(x, loc=x0, scale=gamma)
    np.testing.assert_allclose(ref + sp, -2 * np.log(np.pi * gamma), rtol=1e-12)
    z = (x - x0) / gamma
    np.testing.assert_allclose(ref, np.log1p(z * z) - np.log(np.pi * gamma), rtol=1e-12)
    g = np.load(os.path.join(GOLD, "density_kat.npz"))
    np.testing.assert_allclose(ref, g["cauchy"], rtol=1e-12)


def test_multivariate_log_prob_vs_scipy_and_normaliser_quirk():
    x = np.linspace(-3, 3, 13)
    full = o.multivariate_log_prob(np.full(13, 0.7), 0.2, x, np.float64)
    np.testing.assert_allclose(full, st.norm.logpdf(x, 0.2, 0.7).sum(), rtol=1e-12)
    # Q2: scalar sigma => normaliser counted once (k = size(sigma) = 1)
    scal = o.multivariate_log_prob(0.7, 0.2, x, np.float64)
    one_norm = -0.5 * (2 * np.log(0.7) + np.log(2 * np.pi))
    np.testing.assert_allclose(scal, st.norm.logpdf(x, 0.2, 0.7).sum() - 12 * one_norm, rtol=1e-12)
    g = np.load(os.path.join(GOLD, "density_kat.npz"))
    np.testing.assert_allclose(full, g["mvn_vec"], rtol=1e-12)
    np.testing.assert_allclose(scal, g["mvn_scalar_sigma"], rtol=1e-12)


def test_sigma_clamp():
    assert np.isfinite(o.multivariate_log_prob(0.0, 0.0, np.array([1.0]), np.float32))
    a = o.multivariate_log_prob(1e-12, 0.0, np.array([1e-9]), np.float64)
    b = o.multivariate_log_prob(1e-8, 0.0, np.array([1e-9]), np.float64)
    assert a == b


def test_mvn_diag_scalar_vs_scipy():
    assert abs(o.mvn_diag_scalar_log_prob(0.3, 0.0, 0.2, np.float64) - st.norm.logpdf(0.3, 0, 0.2)) < 1e-12


# ---------------------------------------------------------------- gradients vs torch.autograd (fp64)
def _torch_target(spec, theta, eta, X, Y):
    import torch
    th = torch.tensor(np.asarray(theta, dtype=np.float64), requires_grad=True)
    et = torch.tensor(np.asarray(eta, dtype=np.float64), requires_grad=True)
    Xt, Yt = torch.tensor(X, dtype=torch.float64), torch.tensor(Y, dtype=torch.float64)

    def N(loc, sc):
        return torch.distributions.Normal(torch.tensor(loc, dtype=torch.float64), torch.tensor(sc, dtype=torch.float64))

    def prior(l, h4, W, b, hyper):
        tot = 0
        for x, loc, g in ((W, h4[0], h4[1]), (b, h4[2], h4[3])):
            sc = g ** 2
            if l.prior == o.PRIOR_CAUCHY:
                tot = tot + (torch.log(1 + ((x - loc) / sc) ** 2) - torch.log(math.pi * sc)).sum()
                if hyper:
                    tot = tot + N(0.0, 0.2).log_prob(loc) + \
                        N(0.5 ** 0.5, 0.5).log_prob(sc)
            else:
                s = torch.clamp(sc, 1e-8, 1e8)
                tot = tot - 0.5 * (2 * torch.log(s) + (((x - loc) / s) ** 2).sum() + math.log(2 * math.pi))
                if hyper:
                    tot = tot + N(0.0, 0.1).log_prob(loc) + \
                        N(1.0, 0.1).log_prob(sc)
        return tot

    def lik(f):
        if spec.likelihood == o.LIK_BERNOULLI:
            p = torch.clamp(f, 1e-8, 1 - 1e-7)
            y = Yt.reshape(-1, f.shape[0]).T
            return (torch.xlogy(y, p) + torch.xlogy(1 - y, 1 - p)).sum()
        s = torch.clamp(et[-1] ** 2, 1e-8, 1e8) if spec.likelihood == o.LIK_GAUSSIAN else torch.tensor(spec.fixed_sd, dtype=torch.float64)
        y = Yt.reshape(f.shape[1], -1).T
        n_el = f.numel()
        return -0.5 * (2 * n_el * torch.log(s) + (((y - f) / s) ** 2).sum() + n_el * math.log(2 * math.pi))

    def run(hyper):
        a = Xt.T
        tot = 0
        off = 0
        for i, l in enumerate(spec.layers):
            W = th[off:off + l.in_dim * l.out_dim].reshape(l.out_dim, l.in_dim)
            off += l.in_dim * l.out_dim
            b = th[off:off + l.out_dim].reshape(l.out_dim, 1)
            off += l.out_dim
            tot = tot + prior(l, et[4 * i:4 * i + 4], W, b, hyper)
            z = W @ a + b
            a = {o.ACT_NONE: lambda v: v, o.ACT_RELU: torch.relu, o.ACT_TANH: torch.tanh, o.ACT_SIGMOID: torch.sigmoid, o.ACT_EXP: torch.exp, o.ACT_ELU: torch.nn.functional.elu}[l.act](z)
        if not hyper or spec.likelihood == o.LIK_GAUSSIAN:
            tot = tot + lik(a)
        return tot

    lp = run(False)
    g_th, = torch.autograd.grad(lp, th)
    hl = run(True)
    g_et, = torch.autograd.grad(hl, et, allow_unused=True)
    return float(lp.detach()), g_th.numpy(), float(hl.detach()), g_et.numpy()


CASES = {
    "c1": ([1, 10, 10, 1], 64, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "trainreg": ([1, 10, 10, 10, 1], 11, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN),
    "c2": ([5, 50, 50, 50, 1], 96, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "c5": ([20, 100, 100, 2], 64, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
    "sig": ([4, 7, 3], 50, o.ACT_SIGMOID, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "elu": ([3, 20, 17, 2], 40, o.ACT_ELU, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "exp": ([2, 6, 1], 30, o.ACT_EXP, o.PRIOR_CAUCHY, o.LIK_FIXED_GAUSSIAN),
}


@pytest.mark.parametrize("case", list(CASES))
def test_hand_coded_gradients_match_autograd(case):
    dims, n, act, prior, lik = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    rng = np.random.default_rng(1)
    eta = (eta + 0.05 * rng.standard_normal(eta.size)).astype(np.float32)
    lp_t, g_t, hl_t, hg_t = _torch_target(spec, theta, eta, X, Y)
    lp, g = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp_t) <= 1e-10 * abs(lp_t)
    np.testing.assert_allclose(g, g_t, rtol=1e-9, atol=1e-9 * np.abs(g_t).max())
    hl, hg = o.hyper_log_prob_and_grad(spec, eta, theta, X, Y, np.float64)
    assert abs(hl - hl_t) <= 1e-10 * abs(hl_t)
    np.testing.assert_allclose(hg, hg_t, rtol=1e-8, atol=1e-9 * np.abs(hg_t).max())
    # closed-form data term (sufficient statistic S) == full evaluation
    if spec.likelihood == o.LIK_GAUSSIAN:
        f = o.forward(spec, theta, X, np.float64)
        S = np.sum((Y.reshape(n, -1).T - f) ** 2)
        hl2, hg2 = o.hyper_log_prob_and_grad(spec, eta, theta, X, Y, np.float64, S=S)
        assert abs(hl2 - hl) <= 1e-9 * abs(hl)
        np.testing.assert_allclose(hg2, hg, rtol=1e-9, atol=1e-9 * np.abs(hg).max())


@pytest.mark.parametrize("case", ["c1", "c2"])
def test_float32_arm_within_stated_tolerance(case):
    """the reference's own arithmetic (fp32) sits inside the tolerance band the GPU tests use"""
    dims, n, act, prior, lik = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    lp32, g32 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float32)
    assert abs(lp32 - lp64) <= 4e-6 * abs(lp64)
    assert np.abs(g32 - g64).max() <= 1e-4 * np.abs(g64).max()


# ---------------------------------------------------------------- HMC invariants
def _std_normal_vg(q):
    return -0.5 * float(np.sum(q.astype(np.float64) ** 2)), (-q).astype(q.dtype)


def test_leapfrog_reversibility():
    rng = np.random.default_rng(3)
    q0, p0 = rng.standard_normal(20), rng.standard_normal(20)
    r = o.hmc_step(_std_normal_vg, q0, 0.1, 25, p0, -1e30, np.float64)
    back = o.hmc_step(_std_normal_vg, r.theta_proposed, 0.1, 25, -r.p_final, -1e30, np.float64)
    np.testing.assert_allclose(back.theta_proposed, q0, atol=1e-12)


def test_energy_error_is_second_order():
    rng = np.random.default_rng(4)
    q0, p0 = rng.standard_normal(10), rng.standard_normal(10)
    errs = []
    for eps, L in ((0.1, 10), (0.05, 20), (0.025, 40)):
        errs.append(abs(o.hmc_step(_std_normal_vg, q0, eps, L, p0, 0.0, np.float64).log_accept_ratio))
    assert errs[1] < errs[0] / 3 and errs[2] < errs[1] / 3


def test_small_eps_accepts_and_nonfinite_rejects():
    rng = np.random.default_rng(5)
    q0, p0 = rng.standard_normal(10), rng.standard_normal(10)
    r = o.hmc_step(_std_normal_vg, q0, 1e-4, 5, p0, np.log(0.999), np.float64)
    assert r.accepted and r.accept_prob > 0.999
    bad = lambda q: (float("nan"), np.zeros_like(q))
    r = o.hmc_step(bad, q0, 1e-2, 3, p0, -1e30, np.float64)
    assert not r.accepted and r.log_accept_ratio == -np.inf and r.sjd == 0.0


def test_acceptance_rate_standard_normal():
    """mean accept prob over many transitions on N(0,I) matches exp(min(0,dH)) statistics: > 0.9 at eps=0.2"""
    rng = np.random.default_rng(6)
    q = rng.standard_normal(16)
    acc = []
    for _ in range(300):
        r = o.hmc_step(_std_normal_vg, q, 0.2, 10, rng.standard_normal(16), np.log(rng.random()), np.float64)
        q = r.theta
        acc.append(r.accept_prob)
    assert 0.9 < np.mean(acc) <= 1.0


# ---------------------------------------------------------------- fixtures + C restatement
@pytest.mark.parametrize("name", ["c1", "trainreg", "c2", "c5"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    spec = o.make_spec(list(g["dims"]), int(g["act"]), int(g["prior"]), int(g["lik"]),
                       o.ACT_SIGMOID if int(g["lik"]) == o.LIK_BERNOULLI else o.ACT_NONE)
    lp, gr = o.target_log_prob_and_grad(spec, g["theta"], g["eta"], g["X"], g["Y"], np.float64)
    assert abs(lp - g["logp64"]) <= 1e-12 * abs(g["logp64"])
    np.testing.assert_allclose(gr, g["grad64"], rtol=1e-10, atol=1e-12)
    r = o.weight_step(spec, g["theta"], g["eta"], g["X"], g["Y"], float(g["eps"]), 5, g["p0"], np.log(0.5), np.float64)
    np.testing.assert_allclose(r.trace_logp, g["step_half_trace"], rtol=1e-12)
    assert abs(r.log_accept_ratio - g["step_half_lar"]) < 1e-8
    # fp32 arithmetic (the reference's) gives the same decision within the stated band
    assert abs(g["step_half_lar32"] - g["step_half_lar"]) <= 2e-2 + 1e-4 * abs(g["step_half_lar"])


@pytest.mark.parametrize("name", ["c1", "trainreg", "c2", "c5"])
def test_c_restatement_matches_numpy(name):
    import c_oracle
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    spec = o.make_spec(list(g["dims"]), int(g["act"]), int(g["prior"]), int(g["lik"]),
                       o.ACT_SIGMOID if int(g["lik"]) == o.LIK_BERNOULLI else o.ACT_NONE)
    co = c_oracle.COracle(spec, g["X"], g["Y"])
    lp, gr, _ = co.logp_grad(g["theta"], g["eta"])
    assert abs(lp - g["logp64"]) <= 4e-6 * abs(g["logp64"]) + 1e-3
    assert np.abs(gr - g["grad64"]).max() <= 1e-4 * np.abs(g["grad64"]).max()
    th, acc, lar, lo, ln = co.hmc_step(g["theta"], g["eta"], float(g["eps"]), 5, g["p0"], np.log(0.5))
    assert abs(lar - g["step_half_lar"]) <= 2e-2 + 1e-4 * abs(g["step_half_lar"])
    assert acc == bool(g["step_half_accepted"])


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter/key all zero, and all ones... """
    assert o.philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert o.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    z = o.philox_normals(4000, 50, 0, 1, 0)
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05


def test_dual_averaging_golden():
    g = np.load(os.path.join(GOLD, "c1.npz"))
    stt = o.DualAveragingState(hyper_step_size=0.01, burnin=100)
    for ep, lar in enumerate(g["da_lars"]):
        acc = o.dual_averaging_update(stt, ep, lar)
        np.testing.assert_allclose([acc, stt.h, stt.log_eps_bar, stt.eps_h], g["da_trace"][ep], rtol=1e-6)


def test_sample_files_roundtrip(tmp_path):
    """writer (network.py:545-663) -> reader (predictor.py:43-113)"""
    spec, X, Y, theta, eta = o.synth_problem([1, 10, 10, 1], 8)
    parts = o.unflatten(spec, theta)
    shapes = [s.shape for pr in parts for s in pr]
    w = o.SampleWriter(str(tmp_path / "run"), shapes, ["dense", "relu", "dense", "relu", "dense"], spec.n_hypers,
                       burnin=2, sampling_step=2, networks_per_file=2)
    saved = []
    for it in range(1, 12):
        th = (theta + it).astype(np.float32)
        st_ = [s for pr in o.unflatten(spec, th) for s in pr]
        w.after_epoch(it, st_, [eta[i:i + 1] + it for i in range(eta.size)])
        if it > 2 and it % 2 == 0:
            saved.append(th)
    w.close()
    mats, hyp = o.load_networks(str(tmp_path / "run"))
    n_vis = mats[0].shape[0]          # the last, partially filled file is invisible to the reader
    assert n_vis == 4 and len(hyp) == 4
    for k in range(n_vis):
        got = np.concatenate([m[k].reshape(-1) for m in mats])
        np.testing.assert_allclose(got, saved[k], rtol=1e-6)


def test_hyper_chain_collapses_under_Q1():
    """Why BASELINE configs[4] (Cauchy DenseLayers, hyper-HMC on) has a degenerate hyper chain whatever runs it: with the
    reference's Cauchy 'log-density' (Q1, BNN_functions.py:51-55: + log(1+z^2)) the hyper target of a layer grows like
    -3 cnt log(g^2) as the scale g^2 -> 0, i.e. it is improper with a pole at g = 0.  Driven by the reference's own dual
    averaging (network.py:457-469) the fp64 oracle's chain -- no GPU code involved -- falls onto the pole: the scale of
    the 10,000-weight layer drops by orders of magnitude within 100 epochs and the mean acceptance stays far below the
    0.95 target.  With GaussianDenseLayer priors (a proper target) the same loop settles near the target."""
    res = {}
    for name, prior in (("cauchy", o.PRIOR_CAUCHY), ("gaussian", o.PRIOR_GAUSSIAN)):
        spec, X, Y, theta, eta = o.synth_problem([20, 100, 100, 2], 16, o.ACT_RELU, prior, o.LIK_BERNOULLI)
        # the pole: the target increases monotonically as g_w of layer 1 shrinks (cauchy only)
        vals = []
        for g in (0.7, 0.07, 0.007, 0.0007):
            e = eta.astype(np.float64).copy(); e[5] = g
            vals.append(o.hyper_log_prob(spec, e, theta, X, Y, np.float64))
        res[name + "_monotone"] = all(b > a for a, b in zip(vals, vals[1:]))
        da = o.DualAveragingState(hyper_step_size=1e-2, burnin=10 ** 9)
        rng = np.random.default_rng(99)
        e, acc = eta.astype(np.float64), []
        with np.errstate(all="ignore"):
            for ep in range(100):
                p0 = rng.standard_normal(spec.n_hypers).astype(np.float32)
                r = o.hyper_step(spec, e, theta, X, Y, float(np.float32(da.eps_h)), 100, p0, float(np.log(rng.random())), np.float64)
                e = r.theta.astype(np.float64)
                acc.append(o.dual_averaging_update(da, ep, r.log_accept_ratio))
        res[name] = (abs(e[5]) / eta[5], float(np.mean(acc[50:])))
    assert res["cauchy_monotone"] and not res["gaussian_monotone"]
    assert res["cauchy"][0] < 0.5 and res["cauchy"][1] < 0.5, res          # scale collapsing, acceptance poor
    assert 0.5 < res["gaussian"][0] < 2.0 and res["gaussian"][1] > 0.7, res
