"""GPU: the native RCCL plumbing (tbnn_comm_*, tbnn_gather_samples, tbnn_set_row_shard).

A 1-GPU box can only form a world of one rank (RCCL refuses two ranks on one device), so these
tests pin the plumbing -- librccl.so resolved with dlopen, communicator on the chain's device,
collectives on the chain's stream, the sharded code path (dense-row reduce + all-reduce + separate
statistic buffer) giving the unsharded numbers.  That a sum over row blocks equals the whole is
checked without RCCL in tests/test_gpu_wide.py (row-partition additivity) and below by emulating
two ranks on one GPU with the all-reduce done by hand.
"""
import numpy as np
import pytest

import tbnn_oracle as o
from tensorbnn_amd import parallel

pytestmark = pytest.mark.gpu

SHAPES = {
    "narrow": ([5, 50, 50, 50, 1], 3000, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
    "wide": ([3, 20, 36, 2], 1500, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN),
    "generic": ([4, 9, 2], 700, o.ACT_ELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
}


def chain_of(native, spec):
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_AUTO)


def test_comm_world1_gather(native):
    spec, X, Y, theta, eta = o.synth_problem(*SHAPES["narrow"][:2])
    ch = chain_of(native, spec)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    comm = native.Comm(ch, 1, 0, native.comm_unique_id())
    assert comm.count() == 1                     # ncclCommCount through the real RCCL (what bench.py reports per rank at N > 1)
    g = ch.gather_samples(comm)
    assert g.shape == (1, spec.n_params + spec.n_hypers)
    np.testing.assert_array_equal(g[0, :spec.n_params], theta)
    np.testing.assert_array_equal(g[0, spec.n_params:], eta)
    comm.close(); ch.close()


@pytest.mark.parametrize("shape", list(SHAPES))
def test_row_shard_world1_equals_unsharded(native, shape):
    dims, n, act, prior, lik = SHAPES[shape]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    ref = chain_of(native, spec)
    ref.set_data(X, Y); ref.set_state(theta); ref.set_hypers(eta)
    lp0, g0, st0 = ref.logp_grad(theta, eta)
    out0 = ref.hmc_step(1e-5, 4, p0=p0, log_u=-1e30)
    ch = chain_of(native, spec)
    comm = parallel.make_comm(ch)
    lo, hi = parallel.shard_rows(ch, X, Y, comm)
    assert (lo, hi) == (0, n)
    ch.set_state(theta); ch.set_hypers(eta)
    lp1, g1, st1 = ch.logp_grad(theta, eta)
    out1 = ch.hmc_step(1e-5, 4, p0=p0, log_u=-1e30)
    assert abs(lp1 - lp0) <= 1e-9 * abs(lp0) and abs(st1 - st0) <= 1e-12 * abs(st0)
    np.testing.assert_allclose(g1, g0, rtol=0, atol=2e-6 * np.abs(g0).max())      # same sums, one more reduction level
    assert abs(out1["log_accept_ratio"] - out0["log_accept_ratio"]) <= 1e-3 + 1e-6 * abs(out0["log_accept_ratio"])
    np.testing.assert_allclose(ch.get_state(), ref.get_state(), rtol=0, atol=1e-6 * np.abs(theta).max())
    ch.set_row_shard(None)
    lp2, g2, _ = ch.logp_grad(theta, eta)
    assert lp2 == lp0 and np.array_equal(g2, g0)
    comm.close(); ch.close(); ref.close()


def test_row_shard_two_ranks_emulated(native):
    """two 'ranks' on one GPU, the all-reduce by hand: block gradients minus one prior gradient == full gradient"""
    dims, n, act, prior, lik = SHAPES["narrow"]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    full = chain_of(native, spec); full.set_data(X, Y)
    lp, g, st = full.logp_grad(theta, eta)
    parts = []
    for r in range(2):
        lo, hi = parallel.row_block(n, r, 2)
        assert lo % 16 == 0 and hi > lo
        c = chain_of(native, spec); c.set_data(X[lo:hi], Y[lo:hi])
        parts.append(c.logp_grad(theta, eta)); c.close()
    assert parallel.row_block(n, 1, 2)[1] == n and parallel.row_block(n, 0, 2)[1] == parallel.row_block(n, 1, 2)[0]
    assert abs(parts[0][2] + parts[1][2] - st) <= 1e-6 * abs(st)
    # grad = data_a + data_b + prior; each block call carries the prior once.  The prior gradient from the
    # fp64 oracle: g(rows 0:16) + g(rows 16:32) - g(rows 0:32)
    gg = lambda a, b: o.target_log_prob_and_grad(spec, theta, eta, X[a:b], Y[a:b], np.float64)[1]
    prior_g = gg(0, 16) + gg(16, 32) - gg(0, 32)
    np.testing.assert_allclose(parts[0][1].astype(np.float64) + parts[1][1] - prior_g, g, rtol=0, atol=5e-5 * np.abs(g).max())
    full.close()
