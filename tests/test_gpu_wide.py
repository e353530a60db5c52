"""GPU: the wide-layer path (k_chain_wide + k_dw_wide; configs[3]) and the mid-width fused kernel (k_fwd_bwd_mid;
configs[4]) at BASELINE's full size.

The fp64 oracle cannot evaluate 1e6 x 200-wide rows in seconds, so full-size parity is checked
through size-independent properties: agreement with the (oracle-verified) generic kernel on a
row sample, additivity of the data term over row partitions, determinism, and the leapfrog
energy error shrinking as eps^2.
"""
import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

C4 = ([10, 200, 200, 200, 1], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
C5 = ([20, 100, 100, 2], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI)


def chain(native, spec, kernel):
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    return native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=kernel)


@pytest.mark.parametrize("cfg,n", [(C4, 200_000), (C5, 200_000)], ids=["c4", "c5"])
def test_wide_vs_generic_large(native, cfg, n):
    spec, X, Y, theta, eta = o.synth_problem(cfg[0], n, cfg[1], cfg[2], cfg[3])
    res = {}
    for name, k in (("wide", native.KERNEL_FAST), ("generic", native.KERNEL_GENERIC)):
        ch = chain(native, spec, k)
        if name == "wide":
            assert ch.kernel_name.startswith("wide<" if cfg is C4 else "mid<"), ch.kernel_name
        ch.set_data(X, Y)
        res[name] = ch.logp_grad(theta, eta)
        ch.close()
    lp_w, g_w, st_w = res["wide"]
    lp_g, g_g, st_g = res["generic"]
    assert abs(lp_w - lp_g) <= 1e-6 * abs(lp_g)
    assert np.abs(g_w - g_g).max() <= 5e-5 * np.abs(g_g).max()
    assert abs(st_w - st_g) <= 1e-6 * abs(st_g)


@pytest.mark.parametrize("cfg,n", [(C4, 1_000_000), (C5, 500_000)], ids=["c4_full", "c5_full"])
def test_wide_full_size_properties(native, cfg, n):
    """BASELINE configs[3] / [4] at full size: row-partition additivity + determinism + a short trajectory."""
    spec, X, Y, theta, eta = o.synth_problem(cfg[0], n, cfg[1], cfg[2], cfg[3])
    ch = chain(native, spec, native.KERNEL_FAST)
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp2, g2, st2 = ch.logp_grad(theta, eta)
    assert lp == lp2 and st == st2 and np.array_equal(g, g2)          # deterministic reductions
    # the data term is a sum over rows and the prior is counted once per call: for two different
    # splits of the rows, stat adds up and g_a + g_b (= g + grad prior) is the same vector
    sums = []
    for h in (n // 2 + 7, n // 3 - 5):
        ch.set_data(X[:h], Y[:h]); _, g_a, st_a = ch.logp_grad(theta, eta)
        ch.set_data(X[h:], Y[h:]); _, g_b, st_b = ch.logp_grad(theta, eta)
        assert abs((st_a + st_b) - st) <= 2e-6 * abs(st)
        sums.append(g_a.astype(np.float64) + g_b)
    assert np.abs(sums[0] - sums[1]).max() <= 5e-5 * np.abs(g).max()
    assert np.abs((sums[0] - g) - (sums[1] - g)).max() <= 5e-5 * np.abs(g).max()
    # leapfrog at full size: the energy error of a fixed-length trajectory falls as eps^2
    ch.set_data(X, Y); ch.set_hypers(eta)
    p0 = np.random.default_rng(3).standard_normal(spec.n_params).astype(np.float32)
    T, e1 = (8e-7, 4e-7) if cfg is C4 else (8e-4, 5e-5)
    errs = []
    for eps in (e1, e1 / 2):
        ch.set_state(theta)
        out = ch.hmc_step(eps, int(round(T / eps)), p0=p0, log_u=1e30)      # log u = +inf: never accepted, state kept
        assert np.isfinite(out["log_accept_ratio"]) and out["accepted"] == 0
        errs.append(out["log_accept_ratio"])
    assert 3.0 <= errs[0] / errs[1] <= 5.0, errs
    ch.close()
