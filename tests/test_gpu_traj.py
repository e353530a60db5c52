"""GPU: whole trajectories of small problems in one launch (csrc/kernels_traj.hpp) against the oracle and against the two-kernel step."""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

SHAPES = {
    "configs0": ([1, 10, 10, 1], o.ACT_RELU),              # the ahead-of-time instantiations the trajectory kernel exists for
    "trainreg": ([1, 10, 10, 10, 1], o.ACT_TANH),
    # run-time compiled narrow shapes (the kernel table's `traj` entry): 4 waves per workgroup (their accumulators do not fit four waves per
    # SIMD), fringe units, two outputs, a single hidden layer
    "jit_24": ([6, 24, 24, 1], o.ACT_TANH),
    "jit_fringe_two_outputs": ([7, 17, 33, 2], o.ACT_RELU),
    "jit_one_hidden": ([2, 12, 1], o.ACT_SIGMOID),
}


def make(native, spec, monkeypatch, traj, **kw):
    monkeypatch.setenv("TBNN_TRAJ", "1" if traj else "0")
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True, **kw)


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("n", [7, 100, 380, 1000])     # (16 waves: up to 1,200 rows; 4 waves: up to 384 -- above, both handles take the two-kernel step)
def test_trajectory_kernel_transition_vs_oracle_and_two_kernel_step(native, monkeypatch, shape, n):
    dims, act = SHAPES[shape]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    lp64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[0]
    for L in (1, 2, 9):
        ref = o.weight_step(spec, theta, eta, X, Y, 2e-4, L, p0, -1e30, np.float64)
        got = {}
        for traj in (True, False):
            ch = make(native, spec, monkeypatch, traj)
            ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
            out = ch.hmc_step(2e-4, L, p0=p0, log_u=-1e30)
            got[traj] = (out, ch.get_state())
            assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 4e-7 * abs(lp64), (traj, L)
            assert bool(out["accepted"]) == ref.accepted
            np.testing.assert_allclose(ch.get_state(), ref.theta, rtol=2e-5, atol=2e-6)
            # the state a transition ends in carries its gradient: the next transition starts from it without a bootstrap evaluation
            out2 = ch.hmc_step(2e-4, L, p0=p0, log_u=-1e30)
            ref2 = o.weight_step(spec, ref.theta, eta, X, Y, 2e-4, L, p0, -1e30, np.float64)
            assert abs(out2["log_accept_ratio"] - ref2.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref2.log_accept_ratio) + 4e-7 * abs(lp64), (traj, L)
            ch.close()
        # the two paths sum the gradient in different orders: the same transition to fp32 rounding
        np.testing.assert_allclose(got[True][1], got[False][1], rtol=2e-5, atol=2e-6)


def test_trajectory_kernel_free_running_chain_and_group(native, monkeypatch):
    """on the device's own draws: the production path (tbnn_hmc_run) beside an oracle chain on the same Philox stream; a group of chains at
    their own (eps, L) equals its solo chains bit for bit"""
    dims, act = SHAPES["configs0"]
    n = 1000
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    seed, cid, E, L, eps = 50, 3, 40, 12, 3e-4
    ch = make(native, spec, monkeypatch, True, seed=seed, chain_id=cid)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    recs = ch.hmc_run(eps, L, E)
    th = theta.copy()
    agree = acc = 0
    for e, r in enumerate(recs):
        p0 = o.philox_normals(spec.n_params, seed, cid, e, o.PURPOSE_MOMENTUM)
        lu = float(o.philox_log_uniform(seed, cid, e, o.PURPOSE_LOGU))
        ref = o.weight_step(spec, th, eta, X, Y, eps, L, p0, lu, np.float64)
        margin = abs(ref.log_accept_ratio - lu)
        if margin > 0.05:
            assert bool(r["accepted"]) == ref.accepted, (e, r["log_accept_ratio"], ref.log_accept_ratio, lu)
            agree += 1
        if r["accepted"]:                        # follow the device's decision
            th = o.weight_step(spec, th, eta, X, Y, eps, L, p0, -1e30, np.float64).theta if not ref.accepted else ref.theta
            acc += 1
    assert agree >= E - 6 and 0 < acc
    np.testing.assert_allclose(ch.get_state(), th, rtol=5e-4, atol=5e-5)
    ch.close()
    # group of three at their own (eps, L)
    C = 3
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    monkeypatch.setenv("TBNN_TRAJ", "1")
    thetas = (theta[None, :] * (1.0 + 0.05 * np.random.default_rng(2).standard_normal((C, theta.size)))).astype(np.float32)
    eps_c = np.array([2e-4, 3e-4, 1e-4], dtype=np.float32); L_c = np.array([5, 11, 8], dtype=np.int32)
    grp = native.ChainGroup(layers, C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=7)
    grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(np.tile(eta, (C, 1)))
    g1 = grp.hmc_run_each(eps_c, L_c, 6)
    gs = grp.get_state(); grp.close()
    for c in range(C):
        s = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=7 + c)
        s.set_data(X, Y); s.set_state(thetas[c]); s.set_hypers(eta)
        s1 = s.hmc_run(float(eps_c[c]), int(L_c[c]), 6)
        for a, b in zip(g1[c], s1):
            assert a["log_accept_ratio"] == b["log_accept_ratio"] and a["accepted"] == b["accepted"]
        np.testing.assert_array_equal(gs[c], s.get_state())
        s.close()


def test_profiling_does_not_change_the_path(native, monkeypatch):
    """bench.py switches profiling on after its warm-up; the timed transitions must run on the kernels the warm-up and every user
    run on.  With tbnn_set_profiling the trajectory kernel is still taken, the chain is bit for bit the unprofiled one, and the reported time per
    leapfrog step is the trajectory launch's time / L."""
    dims, act = SHAPES["configs0"]
    spec, X, Y, theta, eta = o.synth_problem(dims, 1000, act, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    runs = {}
    for prof in (0, 3):
        ch = make(native, spec, monkeypatch, True, seed=50, chain_id=1)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        assert ch.last_transition_path == "none"
        ch.set_profiling(prof)
        recs = ch.hmc_run(3e-4, 20, 12)
        assert ch.last_transition_path == "trajectory"
        runs[prof] = ([r["log_accept_ratio"] for r in recs], ch.get_state(), [r["fwdbwd_us"] for r in recs])
        # a traced transition needs the per-step energies: the per-step kernels
        ch.hmc_step(3e-4, 5, trace=True)
        assert ch.last_transition_path == "per-step"
        ch.close()
    assert runs[0][0] == runs[3][0] and np.array_equal(runs[0][1], runs[3][1])
    assert all(u == 0 for u in runs[0][2]) and 0.5 < max(runs[3][2]) < 100.0          # us per leapfrog step of a 1,000-row, 1-10-10-1 trajectory
