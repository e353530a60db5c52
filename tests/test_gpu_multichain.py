"""GPU: several chains behind one handle (tbnn_create_multi, round 4).  Chain c of a group is BIT FOR BIT the chain
tbnn_create(..., chain_id + c) is: same Philox key, same kernels, same summation order -- only that the per-chain kernels of
all chains are one launch each (gridDim.y = chain).  Compared over free-running transitions with accepts and rejects, a hyper
transition in between, on every kernel family (the batched ones and the chain-by-chain ones)."""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

SHAPES = {
    "fast3_c1": ([1, 10, 10, 1], 1000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),            # batched: narrow family (configs[0])
    "fast3_trainreg": ([1, 10, 10, 10, 1], 11, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN),
    "fast_sigmoid": ([4, 7, 3], 130, o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),           # k_fwd_bwd_fast (three outputs on MFMA tiles)
    "mid": ([4, 24, 40, 1], 517, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),                 # batched: mid-width family
    "tall": ([128, 16, 2], 333, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),                    # batched: tall-fan-in family
    "wide": ([3, 20, 36, 2], 517, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),                # chain by chain: two-kernel wide path
    "layered": ([40, 24, 24, 3], 600, o.ACT_TANH, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),              # chain by chain: layered family
    "bern_mid": ([20, 32, 48, 2], 700, o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
}
REC = ("log_accept_ratio", "accepted", "logp_old", "logp_new", "kinetic_old", "kinetic_new", "sjd", "accept_prob")


def layers_of(spec):
    return [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]


@pytest.mark.parametrize("shape", list(SHAPES))
def test_group_chains_are_the_solo_chains(native, shape):
    dims, n, act, prior, lik = SHAPES[shape]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    C, c0, seed = 4, 7, (3 << 32) | 50                         # a high seed word too (folded into the key)
    rng = np.random.default_rng(2)
    thetas = (theta[None, :] * (1.0 + 0.05 * rng.standard_normal((C, theta.size)))).astype(np.float32)
    etas = np.tile(eta, (C, 1)).astype(np.float32) * (1.0 + 0.01 * np.arange(C, dtype=np.float32))[:, None]
    eps_ok, eps_bad = (2e-3 if lik == o.LIK_FIXED_GAUSSIAN else 2e-5), 0.3
    grp = native.ChainGroup(layers_of(spec), C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=c0, jit=False)
    grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(etas)
    g1 = grp.hmc_run(eps_ok, 5, 4)
    g2 = grp.hmc_run(eps_bad, 3, 3)                              # diverging proposals: rejects
    gh = grp.hyper_step(1e-4, 7) if spec.n_hypers else None
    g3 = grp.hmc_step(eps_ok, 4)
    g_state, g_hyp = grp.get_state(), grp.get_hypers()
    kname = grp.kernel_name
    grp.close()
    acc_seen = set()
    for c in range(C):
        ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=c0 + c, jit=False)
        assert ch.kernel_name == kname
        ch.set_data(X, Y); ch.set_state(thetas[c]); ch.set_hypers(etas[c])
        s1 = ch.hmc_run(eps_ok, 5, 4)
        s2 = ch.hmc_run(eps_bad, 3, 3)
        sh = ch.hyper_step(1e-4, 7) if spec.n_hypers else None
        s3 = ch.hmc_step(eps_ok, 4)
        for got, want in list(zip(g1[c] + g2[c], s1 + s2)) + [(g3[c], s3)] + ([(gh[c], sh)] if sh else []):
            np.testing.assert_array_equal(np.array([got[k] for k in REC], dtype=np.float64), np.array([want[k] for k in REC], dtype=np.float64))
            acc_seen.add(int(want["accepted"]))
        np.testing.assert_array_equal(g_state[c], ch.get_state())
        np.testing.assert_array_equal(g_hyp[c], ch.get_hypers())
        ch.close()
    assert acc_seen == {0, 1}, acc_seen                          # both decisions occurred
    assert np.abs(g_state[0] - g_state[1]).max() > 0             # and the chains are different chains


def test_group_guards(native):
    spec, X, Y, theta, eta = o.synth_problem([1, 10, 10, 1], 200)
    grp = native.ChainGroup(layers_of(spec), 3, likelihood=spec.likelihood)
    grp.set_data(X, Y); grp.set_state(theta); grp.set_hypers(eta)
    assert grp.C == 3 and grp.get_state().shape == (3, spec.n_params)
    np.testing.assert_array_equal(grp.get_state(), np.tile(theta, (3, 1)))          # [P] broadcast to every chain
    import ctypes as C
    lp = C.c_double()
    rc = native.lib.tbnn_logp_grad(grp._h, None, None, C.byref(lp), None, None)
    assert rc < 0 and b"multi-chain" in native.lib.tbnn_last_error()
    p0 = np.zeros(spec.n_params, dtype=np.float32)
    out = (native.StepOut * 3)()
    rc = native.lib.tbnn_hmc_step(grp._h, 1e-5, 2, p0.ctypes.data_as(C.POINTER(C.c_float)), None, out, None)
    assert rc < 0 and b"multi-chain" in native.lib.tbnn_last_error()
    with pytest.raises(native.TbnnError):
        native.ChainGroup(layers_of(spec), 0)
    # same seed, same chain ids: identical draws whatever the group size
    a = grp.hmc_run(1e-4, 3, 2)
    grp.close()
    g5 = native.ChainGroup(layers_of(spec), 5, likelihood=spec.likelihood)
    g5.set_data(X, Y); g5.set_state(theta); g5.set_hypers(eta)
    b = g5.hmc_run(1e-4, 3, 2)
    for c in range(3):
        assert [r["log_accept_ratio"] for r in a[c]] == [r["log_accept_ratio"] for r in b[c]]
    g5.close()


@pytest.mark.parametrize("shape", list(SHAPES))
def test_group_chains_at_their_own_step_size_and_leapfrog_count(native, shape):
    """tbnn_hmc_run_each / tbnn_hmc_step_each / tbnn_hyper_step_each: every chain of a group at ITS OWN (eps, L) and hyper step size
    -- the reference runs one adapter per chain (network.py:221-235, :603-607) -- is bit for bit the solo chain driven with those
    values: the lockstep loop runs max L steps, a chain past its own L is skipped (k_update returns, the fused pass's blocks exit)"""
    dims, n, act, prior, lik = SHAPES[shape]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    C, c0, seed = 4, 3, 50
    rng = np.random.default_rng(5)
    thetas = (theta[None, :] * (1.0 + 0.05 * rng.standard_normal((C, theta.size)))).astype(np.float32)
    e0 = 2e-3 if lik == o.LIK_FIXED_GAUSSIAN else 2e-5
    eps = np.array([e0, 0.5 * e0, 2.0 * e0, 0.3], dtype=np.float32)              # (the last one diverges: rejects)
    Ls = np.array([3, 7, 1, 4], dtype=np.int32)
    eps_h = np.array([1e-4, 3e-4, 5e-5, 2e-4], dtype=np.float32)
    grp = native.ChainGroup(layers_of(spec), C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=c0, jit=False)
    grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(eta)
    g1 = grp.hmc_run_each(eps, Ls, 3)
    gh = grp.hyper_step_each(eps_h, 6) if spec.n_hypers else None
    g2 = grp.hmc_step_each(eps[::-1].copy(), Ls[::-1].copy())
    g3 = grp.hmc_step(e0, 2)                                                  # and back to one (eps, L) for all
    g_state, g_hyp = grp.get_state(), grp.get_hypers()
    grp.close()
    for c in range(C):
        ch = native.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=seed, chain_id=c0 + c, jit=False)
        ch.set_data(X, Y); ch.set_state(thetas[c]); ch.set_hypers(eta)
        s1 = ch.hmc_run(float(eps[c]), int(Ls[c]), 3)
        sh = ch.hyper_step(float(eps_h[c]), 6) if spec.n_hypers else None
        s2 = ch.hmc_step(float(eps[C - 1 - c]), int(Ls[C - 1 - c]))
        s3 = ch.hmc_step(e0, 2)
        for got, want in list(zip(g1[c], s1)) + [(g2[c], s2), (g3[c], s3)] + ([(gh[c], sh)] if sh else []):
            np.testing.assert_array_equal(np.array([got[k] for k in REC], dtype=np.float64), np.array([want[k] for k in REC], dtype=np.float64))
            assert got["n_leapfrog"] == want["n_leapfrog"]
        np.testing.assert_array_equal(g_state[c], ch.get_state())
        np.testing.assert_array_equal(g_hyp[c], ch.get_hypers())
        ch.close()


def test_train_chains_flow(tmp_path, monkeypatch, native):
    """network.trainChains: the literal trainRegression problem (11 rows) as 4 chains on the one GPU through the drop-in Python
    API, every chain with its OWN (eps, L) adapter and its own dual averaging of the hyper step size -- as four runs of the
    reference would have (network.py:221-235, :457-469, :603-607).  Chain c of the group IS the single-chain `train` run of
    `network(..., chain_id=c)`: same (eps, L) schedule, same records, bit-identical states after 61 adapted epochs, same samples
    on disk (per-chain folders in the reference's format that `predictor` reads back)."""
    import math
    from tensorbnn_amd.activationFunctions import Tanh
    from tensorbnn_amd.layer import GaussianDenseLayer
    from tensorbnn_amd.likelihood import FixedGaussianLikelihood
    from tensorbnn_amd.network import network
    from tensorbnn_amd.predictor import predictor
    monkeypatch.chdir(tmp_path)
    trainIn = np.linspace(-2, 2, num=11)
    valIn = np.linspace(-2 + 2 / 30, 2.0 - 2 / 30, num=30)
    trainOut = np.sin(trainIn * math.pi * 2) * trainIn - np.cos(trainIn * math.pi)
    valOut = np.sin(valIn * math.pi * 2) * valIn - np.cos(valIn * math.pi)

    def make(chain_id=0):
        net = network(np.float32, 1, trainIn, trainOut.T, valIn, valOut.T, chain_id=chain_id)
        seed = 1000
        net.add(GaussianDenseLayer(1, 10, seed=seed)); net.add(Tanh()); seed += 1000
        for _ in range(2):
            net.add(GaussianDenseLayer(10, 10, seed=seed)); net.add(Tanh()); seed += 1000
        net.add(GaussianDenseLayer(10, 1, seed=seed))
        net.setupMCMC(stepSizeStart=1e-3, stepSizeMin=1e-4, stepSizeMax=1e-2, stepSizeOptions=20, leapfrogStart=50,
                      leapfogMin=10, leapFrogMax=100, leapfrogIncrement=10, hyperStepSize=0.001, hyperLeapfrog=20,
                      burnin=20, averagingSteps=5, randomSteps=3)
        return net

    C, EPOCHS = 4, 61
    rec = make().trainChains(C, EPOCHS, 10, FixedGaussianLikelihood(sd=0.1), adjustHypers=True, folderName="multi", networksPerFile=2)
    assert len(rec) == EPOCHS and all(len(r["main"]) == C for r in rec)
    schedules = set()
    for c in range(C):
        net = make(chain_id=c)
        solo = net.train(EPOCHS, 10, FixedGaussianLikelihood(sd=0.1), adjustHypers=True, folderName="solo%d" % c, networksPerFile=2, verbose=False)
        for rg, rs in zip(rec, solo):                                 # chain c of the group == the single-chain run, epoch by epoch
            assert rg["eps"][c] == rs["eps"] and rg["L"][c] == rs["L"], (c, rg["iter"])
            assert rg["main"][c]["log_accept_ratio"] == rs["main"]["log_accept_ratio"] and rg["main"][c]["accepted"] == rs["main"]["accepted"]
            assert rg["hyper"][c]["log_accept_ratio"] == rs["hyper"]["log_accept_ratio"]
            assert rg["hyper_step_size"][c] == rs["hyper_step_size"]
        schedules.add(tuple((r["eps"][c], r["L"][c]) for r in rec))
        ps = predictor(str(tmp_path / ("solo%d" % c)) + "/")
        p = predictor(str(tmp_path / "multi" / ("chain%d" % c)) + "/")
        assert p.numNetworks == ps.numNetworks == 4 and p.hypers[0].shape == (16,)
        np.testing.assert_array_equal(p.vectors[-1], ps.vectors[-1])          # same samples on disk
        for a, b in zip(p.hypers, ps.hypers):
            np.testing.assert_array_equal(a, b)
    assert len(schedules) == C                                        # the adapters went their own ways
    assert len({r["L"][c] for r in rec for c in range(C)}) > 2         # and moved L


def test_train_chains_initial_states(tmp_path, monkeypatch, native):
    """trainChains(initialStates=[C][P]): over-dispersed starts; a wrong shape is refused"""
    from tensorbnn_amd.activationFunctions import Relu
    from tensorbnn_amd.layer import DenseLayer
    from tensorbnn_amd.likelihood import GaussianLikelihood
    from tensorbnn_amd.network import network
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((200, 1)).astype(np.float32); y = np.sin(x[:, 0]).astype(np.float32)
    net = network(np.float32, 1, x, y, x[:20], y[:20])
    net.add(DenseLayer(1, 10, seed=1)); net.add(Relu()); net.add(DenseLayer(10, 1, seed=2))
    net.setupMCMC(stepSizeStart=1e-4, leapfrogStart=5, burnin=5, adapt=False)
    P = sum(s.size for s in net.states)
    th0 = (net._theta()[None, :] + 0.1 * rng.standard_normal((3, P))).astype(np.float32)
    with pytest.raises(ValueError):
        net.trainChains(3, 2, 1, GaussianLikelihood(sd=0.1), initialStates=th0[:2])
    rec = net.trainChains(3, 4, 1, GaussianLikelihood(sd=0.1), initialStates=th0, adjustHypers=False)
    lp = [rec[0]["main"][c]["logp_old"] for c in range(3)]
    assert len(set(lp)) == 3                                          # three different start states
