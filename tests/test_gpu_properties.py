"""GPU: size-independent properties of the transition at BASELINE's full sizes, on every kernel family -- no oracle involved.

Time reversal: tfp's leapfrog (call sites network.py:394-408; SimpleLeapfrogIntegrator ordering, restated in kernels_hmc.hpp) is
reversible: L steps from (q_0, p_0) reach (q_L, p_L); L steps from (q_L, -p_L) come back to (q_0, -p_0).  On the device that holds up to
fp32 round-off only if every piece is consistent with itself: the half kicks at both ends, the gradient cached for the current state
(quirk Q10: it must be recomputed at q_L, not reused from q_0), the weight images the fused kernels read (scattered by k_update next
to q), the slab reduction.  A forced accept (log u = -1e30) moves the chain to q_L; tbnn_debug_momentum returns p_L.
"""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

#        dims, rows, likelihood, kernel family, eps, L
CASES = {
    "configs[1] narrow": ([5, 50, 50, 50, 1], 100_000, o.LIK_GAUSSIAN, "fast3<", 2e-5, 10),
    "configs[0] narrow": ([1, 10, 10, 1], 1_000, o.LIK_GAUSSIAN, "fast3<", 2e-4, 20),
    "configs[4] mid": ([20, 100, 100, 2], 500_000, o.LIK_BERNOULLI, "mid<", 2e-5, 6),
    "configs[3] wide": ([10, 200, 200, 200, 1], 1_000_000, o.LIK_GAUSSIAN, "wide<", 4e-6, 4),
    "mnist tall": ([784, 20, 20, 1], 12_000, o.LIK_BERNOULLI, "tall<", 2e-4, 8),
    "layered": ([8, 300, 300, 1], 50_000, o.LIK_GAUSSIAN, "layered<", 1e-5, 4),
    # round 6: networks with ten outputs on the three fused families that take them (run-time instantiations; the last layer an MFMA layer)
    "ten outputs wide": ([10, 200, 200, 10], 100_000, o.LIK_BERNOULLI, "jit-wide<", 1e-5, 4),
    "ten outputs mid": ([30, 80, 80, 10], 100_000, o.LIK_BERNOULLI, "jit-mid<", 2e-5, 6),
    "ten outputs tall": ([784, 20, 20, 10], 60_000, o.LIK_BERNOULLI, "jit-tall<", 1e-4, 6),
    # late round 6: shapes that reach their fused kernel through the relaxed limits of jit.families (one hidden layer on the narrow / tall kernels, the wide
    # family behind 100 inputs)
    "one hidden layer narrow": ([1, 100, 1], 100_000, o.LIK_GAUSSIAN, "jit-fast3<", 2e-6, 10),      # (gradients of 1e9 at the initial state: at 1e-4 the kicks are 1e5 x |p_0| and fp32 momenta cannot come back to 4e-3 of it)
    "one hidden layer tall": ([100, 100, 1], 100_000, o.LIK_GAUSSIAN, "jit-tall<", 2e-5, 6),
    "wide behind 100 inputs": ([100, 100, 100, 1], 100_000, o.LIK_GAUSSIAN, "jit-wide", 1e-5, 4),
    # ... and hidden layers with different activations at configs[1]'s size (the packed per-layer activation code)
    "mixed activations narrow": ([5, 50, 50, 50, 1], 100_000, o.LIK_GAUSSIAN, "jit-fast3<tanh+relu+elu", 2e-5, 10),
}
MIXED = {"mixed activations narrow": [o.ACT_TANH, o.ACT_RELU, o.ACT_ELU]}


def _spec_with_acts(case, spec):
    for l, a in zip(spec.layers[:-1], MIXED.get(case, [])):
        l.act = a
    return spec


@pytest.mark.parametrize("case", list(CASES))
def test_time_reversal_full_size(native, case):
    dims, n, lik, family, eps, L = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    spec = _spec_with_acts(case, spec)
    if dims[0] > 100:
        X = (np.abs(X) / 28.0).astype(np.float32)             # pixel-like rows (as tests/test_gpu_tall.py)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=family.startswith("jit-"))
    assert family in ch.kernel_name, ch.kernel_name
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    p0 = np.random.default_rng(11).standard_normal(spec.n_params).astype(np.float32)
    fwd = ch.hmc_step(eps, L, p0=p0, log_u=-1e30, trace=True)
    assert fwd["accepted"] == 1 and np.all(np.isfinite(fwd["trace_logp"]))
    qL, pL = ch.get_state(), ch.debug_momentum()
    moved = float(np.abs(qL - theta).max())
    assert moved > 0 and np.all(np.isfinite(pL))
    back = ch.hmc_step(eps, L, p0=-pL, log_u=-1e30, trace=True)
    assert back["accepted"] == 1
    q_back, p_back = ch.get_state(), ch.debug_momentum()
    # positions: relative to how far the trajectory went; momenta: relative to |p|
    dq = float(np.abs(q_back - theta).max()) / moved
    dp = float(np.abs(p_back + p0).max()) / float(np.abs(p0).max())
    # the log-prob trace is retraced in reverse order, and the two log accept ratios cancel
    tr_f, tr_b = np.asarray(fwd["trace_logp"]), np.asarray(back["trace_logp"])[::-1]
    dtr = float(np.abs(tr_f - tr_b).max() / np.abs(tr_f).max())
    dlar = abs(fwd["log_accept_ratio"] + back["log_accept_ratio"])
    print(f"time reversal {case} on {ch.kernel_name}: |q_back - q_0| / |q_L - q_0| = {dq:.2e}, |p_back + p_0| / |p_0| = {dp:.2e}, "
          f"trace {dtr:.2e}, lar_f + lar_b = {dlar:.2e} (lar_f {fwd['log_accept_ratio']:.4f})")
    ch.close()
    # (measured: dq 2e-7 .. 1.2e-5, dp 4e-7 .. 5.4e-4 -- the trajectories from the initial state see gradients of 1e6 and more --, trace <= 1.7e-7)
    assert dq <= 2e-4 and dp <= 4e-3 and dtr <= 2e-6
    assert dlar <= 4e-2 + 2e-4 * abs(fwd["log_accept_ratio"]) + 2e-6 * abs(tr_f[0])


@pytest.mark.parametrize("case", list(CASES))
def test_row_partition_additivity_full_size(native, case):
    """the data term of the gradient and the likelihood statistic are sums over rows: evaluated over two row blocks (split off a tile
    boundary on purpose) they add up to the evaluation over all rows; the prior's share -- counted once per evaluation -- is taken
    from three small evaluations on the same device path (g[0:16] + g[16:32] - g[0:32]).  Pins the tile loops, the ragged last tile,
    the per-workgroup slabs and their reduction at BASELINE's row counts."""
    dims, n, lik, family, eps, L = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
    spec = _spec_with_acts(case, spec)
    if dims[0] > 100:
        X = (np.abs(X) / 28.0).astype(np.float32)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=family.startswith("jit-"))
    assert family in ch.kernel_name, ch.kernel_name

    def ev(a, b):
        ch.set_data(X[a:b], Y[a:b])
        lp, g, st = ch.logp_grad(theta, eta)
        return lp, g.astype(np.float64), st

    cut = (n * 5 // 8) | 5                                     # not a multiple of 16
    lp_all, g_all, st_all = ev(0, n)
    lp_a, g_a, st_a = ev(0, cut)
    lp_b, g_b, st_b = ev(cut, n)
    g_prior = ev(0, 16)[1] + ev(16, 32)[1] - ev(0, 32)[1]
    ch.close()
    scale = np.abs(g_all).max()
    err = np.abs(g_a + g_b - g_prior - g_all).max() / scale
    dst = abs(st_a + st_b - st_all) / abs(st_all)
    print(f"row additivity {case}: gradient {err:.2e} of its inf-norm, statistic {dst:.2e}")
    assert err <= 2e-6 and dst <= 1e-8            # (measured: 5e-8 .. 1.7e-7 and <= 1.4e-10)
