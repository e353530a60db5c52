"""GPU: run-time instantiated kernels (tensorbnn_amd/jit.py + tbnn_register_kernel_lib) against the oracle."""
import numpy as np
import pytest

import tbnn_oracle as o
from tensorbnn_amd import jit

pytestmark = pytest.mark.gpu

CASES = {
    # narrow family (k_fwd_bwd_fast3): every dW tile in registers
    "narrow_tanh": dict(dims=[6, 24, 24, 1], n=900, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN, name="jit-fast3<"),
    # fast3 with ONE fringe unit per hidden layer (17 = 16 + 1, 33 = 32 + 1: the row-pair form of the B-layout fringe dW)
    # and a 2-output all-fringe last layer; Relu: the packed derivative mask
    "narrow_fringe1": dict(dims=[7, 17, 33, 2], n=700, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, name="jit-fast3<"),
    # fast3, two fringe units, sigmoid hidden layers (the generic derivative path, hardware exp2/reciprocal epilogue)
    "narrow_fringe2_sigmoid": dict(dims=[3, 34, 18, 1], n=600, act=o.ACT_SIGMOID, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN, name="jit-fast3<"),
    # narrow family, 3 outputs on the MFMA path (k_fwd_bwd_fast)
    "narrow_out3": dict(dims=[5, 20, 3], n=500, act=o.ACT_ELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, name="jit-fast<"),
    # mid-width fused kernel (k_fwd_bwd_mid): 5 x 6 + 5 dW tiles in one wave's AccVGPRs
    "mid": dict(dims=[8, 80, 80, 2], n=1200, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI, name="jit-mid<"),
    # ... two middle layers, tanh
    "mid_two_middle": dict(dims=[12, 40, 72, 24, 1], n=1000, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN, name="jit-mid<"),
    # wide family, image resident in LDS (the mid family, which would take this shape first, switched off)
    "wide_resident": dict(dims=[8, 80, 72, 2], n=1200, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI, name="jit-wide(resident)<", skip="mid"),
    # wide family, streamed weights (3 x 128-wide)
    "wide_stream": dict(dims=[8, 128, 128, 128, 1], n=2000, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, name="jit-wide<"),
    # wide family, a middle layer with TWO dW M tiles per wave (114 outputs = 8 tiles) and 12 a-blocks: taken in turn per block the asm
    # accumulators were revisited one MFMA apart and read stale values (dW_2 off by tens of percent, not repeatable; found by
    # tools/experiments/family_fuzz.py) -- the blocks go in pairs now
    "wide_two_m_tiles": dict(dims=[32, 116, 187, 114, 1], n=625, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, name="jit-wide<"),
    # wide family, 6 delta_0 tiles = a group of four + TWO left over on a one-tile input: dW_0's asm form must not take two accumulators in turn
    "wide_dw0_two_left": dict(dims=[8, 96, 72, 2], n=1100, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_GAUSSIAN, name="jit-wide", skip="mid"),
}


@pytest.mark.parametrize("case", list(CASES))
def test_jit_kernel_parity(native, case, monkeypatch):
    c = CASES[case]
    if "skip" in c:
        monkeypatch.setenv("TBNN_JIT_SKIP", c["skip"])
    spec, X, Y, theta, eta = o.synth_problem(c["dims"], c["n"], c["act"], c["prior"], c["lik"])
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_FAST, jit=True)
    assert ch.kernel_name.startswith(c["name"]), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= 4e-6 * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3)
    for _ in range(3):                       # the same launch again: bit-identical (a stale accumulator read is not repeatable)
        lp2, g2, _st = ch.logp_grad(theta, eta)
        assert lp2 == lp and np.array_equal(g2, g)
    # a transition through the registered kernels
    rng = np.random.default_rng(11)
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    ch.set_state(theta); ch.set_hypers(eta)
    out = ch.hmc_step(1e-5, 3, p0=p0, log_u=np.log(0.5))
    ref = o.weight_step(spec, theta, eta, X, Y, 1e-5, 3, p0, np.log(0.5), np.float64)
    # the ratio is a difference of two fp32-path log-probs: its error scales with their magnitude
    assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 2e-7 * abs(lp64)
    ch.close()


@pytest.mark.parametrize("case,grid", [("narrow_fringe1", 5), ("narrow_fringe2_sigmoid", 4), ("narrow_tanh", 7)])
def test_jit_kernel_cooperative_tail(native, monkeypatch, case, grid):
    """the cooperative tail (Coop3) of run-time instantiated narrow kernels: one / two fringe units per layer, a 2-output last
    layer, non-Relu hidden layers; a small grid (TBNN_FAST_GRID) puts these row counts into the rounds + left-over regime"""
    c = CASES[case]
    spec, X, Y, theta, eta = o.synth_problem(c["dims"], c["n"], c["act"], c["prior"], c["lik"])
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ntiles, W = (c["n"] + 15) // 16, 4 * grid
    assert 0 < ntiles % W <= 2 * grid, "the case must leave a cooperative remainder"
    monkeypatch.setenv("TBNN_FAST_GRID", str(grid))
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_FAST, jit=True)
    assert ch.kernel_name.startswith("jit-fast3<"), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= 4e-6 * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (case, a, b)
    ch.close()


# shapes with >= 2048 parameters whose staging leaves room for the dense slab copy: the wave-specialised epilogue (Epi3: three staged
# copies per tile, the own copy in registers, dense slab through LDS, 16-byte stores) in its variants -- two / one / no fringe unit
# per hidden layer in the B-operand and the D layout, N-fringe accumulator pairs, a 2-output last layer -- each with no, one and two
# cooperative tiles per workgroup (their accumulators are merged while staging)
EPI_CASES = {
    "fringe2": ([8, 50, 50, 1], o.ACT_RELU),
    "fringe1_three_hidden": ([5, 49, 49, 49, 1], o.ACT_TANH),
    "no_fringe_d_layout": ([6, 51, 51, 1], o.ACT_RELU),
    "two_outputs": ([9, 35, 35, 35, 2], o.ACT_RELU),
    "mixed_widths": ([12, 50, 40, 1], o.ACT_ELU),
    "four_hidden": ([3, 33, 33, 33, 33, 1], o.ACT_RELU),
    "wide_input": ([16, 49, 51, 1], o.ACT_SIGMOID),
    # five layers, a 16-wide and a 32-wide one (ones slots in tiles of their own), a fringe unit: the shape on which the two-copy cooperative
    # tail lost the second tile's share of a bias column (narrow_fuzz.py; cured by a barrier between the copies, cause not isolated)
    "five_layers_two_tiles": ([13, 36, 16, 33, 32, 2], o.ACT_SIGMOID),
}


@pytest.fixture(scope="module")
def epi_kernels():
    """the seven kernel libraries compiled side by side (one hipcc each, ~1 min) instead of one after the other"""
    from concurrent.futures import ThreadPoolExecutor
    def one(case):
        dims, act = EPI_CASES[case]
        spec = o.make_spec(dims, act, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN, o.ACT_NONE)
        return jit.build([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], spec.likelihood)
    with ThreadPoolExecutor(max_workers=len(EPI_CASES)) as ex:
        paths = list(ex.map(one, EPI_CASES))
    assert all(paths), paths
    return paths


@pytest.mark.parametrize("case", list(EPI_CASES))
@pytest.mark.parametrize("ncoop", [0, 1, 2])
def test_jit_narrow_epilogue_variants(native, monkeypatch, epi_kernels, case, ncoop):
    dims, act = EPI_CASES[case]
    grid = 6
    W = 4 * grid
    n = 16 * (2 * W + (0, 3, grid + 2)[ncoop]) - 5                # two full rounds + 0 / 3 / grid + 2 left-over tiles, a ragged last tile
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    assert spec.n_params >= 2048
    monkeypatch.setenv("TBNN_FAST_GRID", str(grid))
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_FAST, jit=True)
    assert ch.kernel_name.startswith("jit-fast3<"), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= 4e-6 * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (case, ncoop, a, b)
    ch.close()


@pytest.mark.parametrize("n", [582, 630])
def test_jit_second_cooperative_tile_keeps_every_register(native, monkeypatch, epi_kernels, n):
    """13 -> 36 -> 16 -> 33 -> 32 -> 2 over a grid of 4 workgroups with 5 / 8 left-over tiles (one / all workgroups run TWO cooperative tiles): the
    configuration narrow_fuzz.py found (dW_2's bias column off by 4 % / 19 % in registers 2, 3 of every lane group before the barrier between
    the two copies of the tile)"""
    dims, act = EPI_CASES["five_layers_two_tiles"]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, o.PRIOR_GAUSSIAN, o.LIK_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    monkeypatch.setenv("TBNN_FAST_GRID", "4")
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, kernel=native.KERNEL_FAST, jit=True)
    assert ch.kernel_name.startswith("jit-fast3<"), ch.kernel_name
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
    assert abs(lp - lp64) <= 4e-6 * abs(lp64) + 1e-3
    for l, (ow, ob) in zip(spec.layers, spec.offsets()):
        for a, b in ((ow, ob), (ob, ob + l.out_dim)):
            assert np.abs(g[a:b] - g64[a:b]).max() <= 1e-4 * max(np.abs(g64[a:b]).max(), 1e-3), (n, a, b)
    ch.close()


def test_jit_unsupported_shape_falls_back(native):
    """no fused family for this stack (ten hidden layers with alternating activations: the packed per-layer code holds nine; mixed activations
    as such run on the fused kernels since round 6, tests/test_gpu_mixedact.py) -> AUTO runs on the layered MFMA kernels, FAST fails loudly"""
    acts = [native.ACT_RELU, native.ACT_TANH] * 5
    layers = [(4, 8, acts[0], native.PRIOR_CAUCHY)] + [(8, 8, a, native.PRIOR_CAUCHY) for a in acts[1:]] + [(8, 1, native.ACT_NONE, native.PRIOR_CAUCHY)]
    assert jit.shape_of(layers, native.LIK_GAUSSIAN) is None
    ch = native.Chain(layers, kernel=native.KERNEL_AUTO, jit=True)
    assert ch.kernel_name == "layered<4," + "8," * 10 + "1>"
    ch.close()
    with pytest.raises(native.TbnnError):
        native.Chain(layers, kernel=native.KERNEL_FAST, jit=True)
