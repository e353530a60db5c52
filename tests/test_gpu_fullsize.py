"""GPU parity at BASELINE.json's full sizes against the C restatement of the reference's path (oracle/c), and
free-running chains (no per-epoch state reset) against the oracles with the same injected draws.

  * configs[3] (10->200->200->200->1, n = 1e6) and configs[4] (20->100->100->2 Bernoulli, n = 5e5): one value +
    gradient and one short injected transition on the wide path, as tests/test_gpu_parity.py does for configs[1];
  * configs[1] (n = 1e5): >= 200 free-running epochs (L = 10) from the burned-in state at the fixture's step size:
    per-epoch log accept ratio, decision agreement, and the north-star's accept-ratio parity of +-0.02;
  * configs[4]'s parameter count (P = 12,402): 100 free-running hyper transitions (L_h = 100) with the reference's
    dual averaging driving the step size, eta trajectory and per-epoch log accept ratio against the fp64 oracle.

Injected uniforms: log u is drawn from the seeded stream and redrawn while it lies within 0.05 of the ORACLE's log
accept ratio, so that a Metropolis decision never hinges on the last bits of two fp32 sums; a decision mismatch then
means a log-accept-ratio disagreement above the stated tolerance (2e-2 + 1e-4 |lar|), i.e. a real defect.
"""
import os

import numpy as np
import pytest

import c_oracle
import tbnn_oracle as o

pytestmark = pytest.mark.gpu

LOGP_RTOL = 4e-6
GRAD_RTOL = 1e-4


def lar_tol(lar, logp):
    """stated fp32 tolerance of a log accept ratio: 2e-2 + 1e-4 |lar| (tests/test_gpu_parity.py) + 1e-6 |logp| -- lar is a
    difference of two fp32-evaluated log-probs, each good to ~1e-6 relative (LOGP_RTOL is 4e-6); at the burned-in
    configs[1] state |logp| = 1.06e5, so 0.1 of absolute slack is the resolution of the quantity itself"""
    return 2e-2 + 1e-4 * abs(lar) + 1e-6 * abs(logp)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def layers_of(spec):
    return [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]


def check_full_size(native, dims, n, lik, eps, L, family):
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, jit=family.startswith("jit-"))
    assert ch.kernel_name.startswith(family), ch.kernel_name
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    co = c_oracle.COracle(spec, X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp_c, g_c, st_c = co.logp_grad(theta, eta)
    assert abs(lp - lp_c) <= LOGP_RTOL * abs(lp_c), (lp, lp_c)
    assert abs(st - st_c) <= LOGP_RTOL * abs(st_c), (st, st_c)
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            err = np.abs(g[a:b] - g_c[a:b]).max()
            assert err <= GRAD_RTOL * np.abs(g_c[a:b]).max(), (a, b, err, np.abs(g_c[a:b]).max())
    rng = np.random.default_rng(77)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    lu = float(np.log(0.3))
    out = ch.hmc_step(eps, L, p0=p0, log_u=lu, trace=True)
    th_c, acc_c, lar_c, lp_old, lp_new = co.hmc_step(theta, eta, eps, L, p0, lu)
    assert abs(out["logp_old"] - lp_old) <= LOGP_RTOL * abs(lp_old)
    assert abs(out["trace_logp"][0] - lp_old) <= LOGP_RTOL * abs(lp_old)
    assert abs(out["logp_new"] - lp_new) <= LOGP_RTOL * abs(lp_new)
    assert abs(out["trace_logp"][-1] - lp_new) <= LOGP_RTOL * abs(lp_new)
    assert abs(out["log_accept_ratio"] - lar_c) <= 2e-2 + 1e-4 * abs(lar_c), (out["log_accept_ratio"], lar_c)
    assert bool(out["accepted"]) == acc_c
    if acc_c:
        np.testing.assert_allclose(ch.get_state(), th_c, rtol=0, atol=2e-6 * np.abs(th_c).max() + 1e-7)
    ch.close()


def test_configs3_full_size_vs_c_restatement(native):
    """BASELINE configs[3]: 10->200->200->200->1, n = 1e6 on k_chain_wide + k_dw_wide"""
    check_full_size(native, [10, 200, 200, 200, 1], 1_000_000, o.LIK_GAUSSIAN, eps=1e-6, L=2, family="wide<")


def test_ten_outputs_on_the_wide_family_full_size_vs_c_restatement(native):
    """round 6: 10->200->200->10 + Sigmoid, BernoulliLikelihood over ten outputs, n = 1e5 (bench workload wm10) on k_chain_wide + k_dw_wide with the last
    layer as one more streamed middle layer -- value, statistic, gradient per tensor and a three-step transition against oracle/c"""
    check_full_size(native, [10, 200, 200, 10], 100_000, o.LIK_BERNOULLI, eps=2e-5, L=3, family="jit-wide<")


def test_configs4_full_size_vs_c_restatement(native):
    """BASELINE configs[4]: 20->100->100->2 + Sigmoid, BernoulliLikelihood, n = 5e5 on the wide path"""
    check_full_size(native, [20, 100, 100, 2], 500_000, o.LIK_BERNOULLI, eps=5e-5, L=3, family="mid<")


def away_from(rng, lar, margin=0.05):
    """log U(0,1) from the seeded stream, redrawn while within `margin` of lar"""
    while True:
        lu = float(np.log(rng.random()))
        if abs(lu - lar) >= margin:
            return lu


def test_free_running_accept_ratio_parity_configs1(native):
    """north_star: "accept-ratio parity +-0.02" -- a free-running configs[1] chain, n = 1e5, 200 epochs of L = 10 from the
    burned-in state at the fixture's step size; the oracle/c chain and the HIP chain receive the same p0 and log u every
    epoch and each carries its OWN state forward (no reset)."""
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 100_000)
    fx = os.path.join(GOLDEN, "c2_burned.npz")
    if os.path.exists(fx):
        z = np.load(fx)
        theta, eta, eps = z["theta"].astype(np.float32), z["eta"].astype(np.float32), float(z["eps"])
    else:                                            # before the fixture exists: the initial state, its stable step size
        eps = 2e-5
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood)
    assert ch.kernel_name.startswith("fast3<")
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    co = c_oracle.COracle(spec, X, Y)
    rng = np.random.default_rng(2024)
    EPOCHS, L = 200, 10
    th_c = theta.copy()
    acc_g, acc_c, dlar, agree, state_err = [], [], [], 0, []
    for ep in range(EPOCHS):
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        q_c, lar_c, lp0_c, _ = co.hmc_propose(th_c, eta, eps, L, p0)
        lu = away_from(rng, lar_c)
        if lu < lar_c:
            th_c = q_c
        out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
        agree += int(bool(out["accepted"]) == (lu < lar_c))
        dlar.append(abs(out["log_accept_ratio"] - lar_c) / lar_tol(lar_c, lp0_c))         # in units of the stated tolerance
        acc_g.append(out["accept_prob"]); acc_c.append(min(1.0, float(np.exp(min(lar_c, 0.0)))))
        if ep % 20 == 19:
            state_err.append(float(np.abs(ch.get_state() - th_c).max() / np.abs(th_c).max()))
    mg, mc = float(np.mean(acc_g)), float(np.mean(acc_c))
    print(f"free-running configs[1]: accept ratio HIP {mg:.4f} oracle/c {mc:.4f}; decision agreement {agree}/{EPOCHS}; "
          f"max |dlar| / lar_tol {max(dlar):.3f}; relative state distance every 20 epochs {['%.1e' % s for s in state_err]}")
    assert abs(mg - mc) <= 0.02, (mg, mc)
    assert agree == EPOCHS, (agree, max(dlar))
    assert max(dlar) <= 1.0, max(dlar)
    if os.path.exists(fx):
        assert 0.55 <= mc <= 0.95, mc              # the regime SURVEY 8(d) asks for
    ch.close()


def test_log_accept_ratio_band_against_fp64(native):
    """Which arm is off when a burned-in log accept ratio differs between the HIP path and oracle/c?  Ten free-running
    configs[1] epochs (n = 1e5, L = 10) from the burned-in state; every epoch's trajectory is also integrated by the fp64
    NumPy oracle from the HIP chain's own start state, and both fp32 arms are compared with it SEPARATELY against the stated
    band 2e-2 + 1e-4 |lar| (BASELINE.md section 5 adds 1e-6 |logp| -- lar is a difference of two fp32-evaluated log-probs of
    magnitude 1e5 -- which these epochs do not need)."""
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 100_000)
    z = np.load(os.path.join(GOLDEN, "c2_burned.npz"))
    theta, eta, eps = z["theta"].astype(np.float32), z["eta"].astype(np.float32), float(z["eps"])
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    co = c_oracle.COracle(spec, X, Y)
    rng = np.random.default_rng(31)
    L, rows = 10, []
    for ep in range(10):
        th = ch.get_state()
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        ref = o.weight_step(spec, th, eta, X, Y, eps, L, p0, 0.0, np.float64)
        _, lar_c, lp0_c, _ = co.hmc_propose(th, eta, eps, L, p0)
        lu = away_from(rng, ref.log_accept_ratio)
        out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
        assert bool(out["accepted"]) == (lu < ref.log_accept_ratio)
        rows.append((ref.log_accept_ratio, out["log_accept_ratio"] - ref.log_accept_ratio, lar_c - ref.log_accept_ratio, lp0_c))
    rows = np.array(rows)
    band = 2e-2 + 1e-4 * np.abs(rows[:, 0])
    print("lar (fp64)            :", np.array2string(rows[:, 0], precision=3))
    print("HIP - fp64            :", np.array2string(rows[:, 1], precision=4), "max / band", float(np.max(np.abs(rows[:, 1]) / band)))
    print("oracle/c - fp64       :", np.array2string(rows[:, 2], precision=4), "max / band", float(np.max(np.abs(rows[:, 2]) / band)))
    # both fp32 arms inside the BARE band 2e-2 + 1e-4 |lar| (no |logp| term).  Until round 5 the synthetic targets came out of an fp32
    # NumPy matmul whose last bits followed the BLAS thread count, and on one realisation an epoch of the HIP arm reached 1.36 x the band;
    # with the targets synthesised in fp64 and rounded once (oracle.synth_problem) the worst of these ten epochs is 0.36 x the band.
    assert np.all(np.abs(rows[:, 1]) <= band), rows
    assert np.all(np.abs(rows[:, 2]) <= band), rows
    ch.close()


def free_running_wide(native, name, dims, n, lik, epochs, L, family):
    """free-running chain at the burned-in state of a wide config against oracle/c: same p0 / log u, each arm carries its own state"""
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    z = np.load(os.path.join(GOLDEN, f"{name}_burned.npz"))
    theta, eta, eps = z["theta"].astype(np.float32), z["eta"].astype(np.float32), float(z["eps"])
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood)
    assert ch.kernel_name.startswith(family), ch.kernel_name
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    co = c_oracle.COracle(spec, X, Y)
    rng = np.random.default_rng(77)
    th_c = theta.copy()
    acc_g, acc_c, dlar, agree = [], [], [], 0
    for ep in range(epochs):
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        q_c, lar_c, lp0_c, _ = co.hmc_propose(th_c, eta, eps, L, p0)
        lu = away_from(rng, lar_c)
        if lu < lar_c:
            th_c = q_c
        out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
        agree += int(bool(out["accepted"]) == (lu < lar_c))
        dlar.append(abs(out["log_accept_ratio"] - lar_c) / lar_tol(lar_c, lp0_c))
        acc_g.append(out["accept_prob"]); acc_c.append(min(1.0, float(np.exp(min(lar_c, 0.0)))))
    mg, mc = float(np.mean(acc_g)), float(np.mean(acc_c))
    err = float(np.abs(ch.get_state() - th_c).max() / np.abs(th_c).max())
    print(f"free-running {name}: accept ratio HIP {mg:.4f} oracle/c {mc:.4f}; decisions {agree}/{epochs}; max |dlar| / lar_tol {max(dlar):.3f}; "
          f"relative state distance at the end {err:.1e}")
    assert abs(mg - mc) <= 0.02, (mg, mc)
    assert agree == epochs, (agree, max(dlar))
    assert max(dlar) <= 1.0, max(dlar)
    assert err <= 1e-4
    ch.close()


def test_free_running_configs4_burned_in(native):
    """BASELINE configs[4] (20->100->100->2 Bernoulli, n = 5e5) on the kernel bench.py times (k_fwd_bwd_mid) at the state it
    times: 50 free-running epochs of L = 10 from the burned-in fixture at its step size, against oracle/c"""
    free_running_wide(native, "c5", [20, 100, 100, 2], 500_000, o.LIK_BERNOULLI, epochs=50, L=10, family="mid<")


def test_free_running_configs3_burned_in(native):
    """BASELINE configs[3] (10->200->200->200->1, n = 1e6) on k_chain_wide + k_dw_wide at the burned-in state: 10 free-running
    epochs of L = 5 at the fixture's step size, against oracle/c"""
    free_running_wide(native, "c4", [10, 200, 200, 200, 1], 1_000_000, o.LIK_GAUSSIAN, epochs=10, L=5, family="wide<")


@pytest.mark.parametrize("prior", ["cauchy", "gaussian"])
def test_free_running_hyper_chain_full_parameter_count(native, prior):
    """configs[4]'s P = 12,402 (20->100->100->2): 100 free-running hyper transitions of L_h = 100 with the step size driven by
    the reference's dual averaging (network.py:457-469) from setupMCMC's default 1e-2 (Q6: mu = log(100 eps0), so the
    adaptation starts near 1 and needs ~40 epochs to come down to a stable size).
      cauchy   -- configs[4] as BASELINE has it (DenseLayer = CauchyDenseLayer).  Under the reference's Cauchy
                  'log-density' (Q1: + log(1+z^2)) the hyper target is IMPROPER: it grows without bound as a scale g^2 -> 0.
                  eta falls onto that pole faster than the adaptation shrinks the step, most proposals blow up and are
                  rejected; a FIXED step (round 1's bench) ends at a 0.0 accept ratio.  The fp64 oracle alone shows the same
                  (tests/test_oracle.py::test_hyper_chain_collapses_under_Q1), so this is the reference's target, not the
                  sampler: asserted here is that the HIP arm follows the oracle through it, decision by decision.
      gaussian -- GaussianDenseLayer priors (layer.py:282-459): a proper target; the same kernel is a healthy sampler
                  (accept ratio on its way to the 0.95 target, eta moving).
    Bernoulli: the hyper target has no data term, so a few rows do."""
    from tensorbnn_amd.network import DualAveraging
    pk = o.PRIOR_CAUCHY if prior == "cauchy" else o.PRIOR_GAUSSIAN
    spec, X, Y, theta, eta = o.synth_problem([20, 100, 100, 2], 64, o.ACT_RELU, pk, o.LIK_BERNOULLI)
    assert spec.n_params == 12402 and spec.n_hypers == 12
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    da_g, da_o = DualAveraging(1e-2, 10 ** 9), o.DualAveragingState(hyper_step_size=1e-2, burnin=10 ** 9)
    rng = np.random.default_rng(99)
    e_o = eta.astype(np.float64)
    accs, n_acc, worst, blown = [], 0, 0.0, 0
    # gaussian: the stated tolerance.  cauchy: the chain runs down the pole of an improper target, where the trajectories
    # are stiff -- the fp32 arm of the ORACLE differs from its fp64 arm by 0.03 at |lar| = 1.7 there (measured), so the
    # band is that of the arithmetic, not of the implementation
    tol_a, tol_r, eta_rtol = (2e-2, 1e-3, 2e-4) if prior == "gaussian" else (1e-1, 3e-2, 5e-3)
    with np.errstate(all="ignore"):                   # blown-up trajectories overflow in the oracle too
        for ep in range(100):
            p0 = rng.standard_normal(spec.n_hypers).astype(np.float32)
            # the HIP arm's own adaptation sets the step size of the epoch (both arms integrate with it); the oracle's
            # adaptation, fed with the oracle's log accept ratios, must track it
            eps_o, eps_g = float(np.float32(da_o.eps_h)), float(da_g.step_size)
            assert abs(eps_g - eps_o) <= 2e-2 * eps_o, (ep, eps_g, eps_o)
            ref = o.hyper_step(spec, e_o, theta, X, Y, eps_g, 100, p0, 1e30, np.float64)      # log u = +inf: the proposal only
            lar_o = ref.log_accept_ratio
            if not (abs(lar_o) < 50.0):
                # an unstable trajectory (the step of the moment is far too large: energy errors of 1e2 .. inf, of either
                # sign when it lands next to the pole of the improper Cauchy target).  Its value
                # is not comparable digit by digit -- near the pole of the improper Cauchy target even its SIGN depends on
                # the arithmetic: at epoch 40 of this very sequence the fp64 oracle gets -3.1e4, the fp32 oracle arm
                # +3.8e3 and the HIP path +5.5e3 (a blown-up trajectory that lands next to the pole).  Both arms reject
                # it through the injected uniform; everything else is compared.
                lu = 1e30
                out = ch.hyper_step(eps_g, 100, p0=p0, log_u=lu)
                assert not out["accepted"]
                blown += 1
            else:
                lu = away_from(rng, lar_o, 0.05 if prior == "gaussian" else 0.25)
                out = ch.hyper_step(eps_g, 100, p0=p0, log_u=lu)
                assert abs(out["log_accept_ratio"] - lar_o) <= tol_a + tol_r * abs(lar_o), (ep, out["log_accept_ratio"], lar_o)
                worst = max(worst, abs(out["log_accept_ratio"] - lar_o))
            acc_o = lu < lar_o
            assert bool(out["accepted"]) == acc_o, (ep, out["log_accept_ratio"], lar_o, lu)
            if acc_o:
                e_o = ref.theta_proposed.astype(np.float64)
                n_acc += 1
            np.testing.assert_allclose(ch.get_hypers(), e_o, rtol=eta_rtol, atol=2e-6, err_msg=f"eta after epoch {ep}")
            # both arms adapt on their OWN log accept ratio (a blown-up trajectory counts as the rejection it was made)
            accs.append(float(da_g.update(ep, out["log_accept_ratio"] if abs(lar_o) < 50.0 else lar_o)))
            o.dual_averaging_update(da_o, ep, lar_o)
    print(f"hyper chain [{prior}]: accepted {n_acc}/100, mean accept prob {np.mean(accs):.3f} (last 50: {np.mean(accs[50:]):.3f}), "
          f"step size 1e-2 -> {float(da_g.step_size):.3e}, {blown} blown-up trajectories (both arms reject), max |dlar| of the "
          f"others {worst:.2e}, g_w of layer 1: {eta[5]:.4f} -> {e_o[5]:.4e}")
    if prior == "gaussian":
        assert n_acc >= 40 and np.mean(accs[50:]) > 0.6       # a real sampler: the adapted step size keeps accepting
        assert abs(e_o[5] - eta[5]) > 1e-3                    # and eta moved
    else:
        assert abs(e_o[5]) < 0.5 * eta[5]                     # on its way to the pole at g = 0, in both arms
    ch.close()


def test_trace_over_several_steps_with_a_reject(native):
    """tbnn_hmc_step(trace) from a CACHED current state: after a rejected transition (and after a tbnn_logp_grad probe) the
    statistic buffer holds another point's value; logp_old / trace[0] / the next log accept ratio must still be those of
    the current state"""
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 1000)
    ch = native.Chain(layers_of(spec), likelihood=spec.likelihood)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    rng = np.random.default_rng(4)
    th = theta.copy()
    for k, lu in enumerate([np.log(0.5), 1e30, np.log(0.5), 1e30, 1e30, np.log(0.9)]):      # 1e30: forced reject
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        if k == 2:                                   # a probe at another point between two transitions
            ch.logp_grad((th * 1.01).astype(np.float32), eta)
        out = ch.hmc_step(5e-5, 4, p0=p0, log_u=float(lu), trace=True)
        ref = o.weight_step(spec, th, eta, X, Y, 5e-5, 4, p0, float(lu), np.float64)
        np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=LOGP_RTOL, atol=2e-3, err_msg=f"step {k}")
        assert abs(out["logp_old"] - ref.logp_old) <= LOGP_RTOL * abs(ref.logp_old) + 2e-3, k
        assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= lar_tol(ref.log_accept_ratio, ref.logp_old), k
        assert bool(out["accepted"]) == ref.accepted, k
        th = ch.get_state()
        np.testing.assert_allclose(th, ref.theta, rtol=0, atol=2e-5 * max(1.0, np.abs(ref.theta).max()))
    # the hyper transition reads the cached statistic of the CURRENT state too
    lp, g = ch.hyper_logp_grad(eta)
    lp64, g64 = o.hyper_log_prob_and_grad(spec, eta, th, X, Y, np.float64)
    assert abs(lp - lp64) <= LOGP_RTOL * abs(lp64) + 1e-3
    ch.close()
