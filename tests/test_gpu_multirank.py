"""GPU: the N > 1 data path of the C ABI, executed -- two processes on the one GPU of the test box running
tbnn_comm_create(world=2), tbnn_gather_samples and the row-sharded chain (tbnn_set_row_shard) end to end.

RCCL refuses two ranks on one device, so libtbnn's collective library (resolved with dlopen) is pointed at
tests/stubccl (shared-memory all-gather / all-reduce between processes, test infrastructure) through
TBNN_RCCL_LIB.  Everything else is the product path: per-rank chains, dense-row reduce, all-reduce of the gradient
row and of the statistic buffer on the chain's stream, k_update / k_energy / k_hyper on the reduced values.
Asserted: the gather is chain-major and identical on every rank; the sharded chain takes the unsharded chain's
trajectory (same log-prob trace, log accept ratio, decision, state, hyper step) on the narrow, wide and generic
kernels, Gaussian and Bernoulli likelihoods; and the unsharded chain is the oracle's (tests/test_gpu_parity.py).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
STUB_DIR = os.path.join(HERE, "stubccl")
STUB = os.path.join(STUB_DIR, "libstubccl.so")


def build_stub():
    src = os.path.join(STUB_DIR, "stub_ccl.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", STUB, src, "-lrt"])
    return STUB


def run_world(mode, world, tmp_path):
    from conftest import wait_gpu_quiet
    wait_gpu_quiet()
    env = dict(os.environ, TBNN_RCCL_LIB=build_stub(), TBNN_JIT="0")
    idfile = str(tmp_path / f"{mode}.id")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"{mode}_{r}.npz")
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(STUB_DIR, "worker.py"), mode, str(r), str(world), idfile, out],
                                            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    res = []
    for out, p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()
            raise
        assert p.returncode == 0, log[-3000:]
        res.append(np.load(out))
    return res


def test_gather_two_ranks(tmp_path):
    r0, r1 = run_world("gather", 2, tmp_path)
    for it in range(3):
        g0, g1 = r0[f"g{it}"], r1[f"g{it}"]
        assert g0.shape == (2, 5451 + 17)
        np.testing.assert_array_equal(g0, g1)                           # every rank ends up with all samples
        np.testing.assert_array_equal(g0[0], r0[f"own{it}"])            # chain-major: row r = rank r's (theta, eta)
        np.testing.assert_array_equal(g0[1], r1[f"own{it}"])
    assert np.abs(r0["own2"] - r1["own2"]).max() > 0.5                  # the two chains really are different chains


def test_gather_four_ranks(tmp_path):
    """the same at world = 4 (the GPU box admits six processes on its card: this one and four ranks leave one to spare)"""
    res = run_world("gather", 4, tmp_path)
    for it in range(3):
        g0 = res[0][f"g{it}"]
        assert g0.shape == (4, 5451 + 17)
        for r in range(4):
            np.testing.assert_array_equal(res[r][f"g{it}"], g0)
            np.testing.assert_array_equal(g0[r], res[r][f"own{it}"])
    assert all(int(r["comm_count"]) == 4 for r in res)


def test_row_shard_two_ranks(tmp_path, native):
    import worker                                                       # shapes only
    ranks = run_world("shard", 2, tmp_path)
    for name, (dims, n, act, prior, lik) in worker.SHAPES.items():
        spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
        ref = worker.chain_of(spec)
        ref.set_data(X, Y); ref.set_state(theta); ref.set_hypers(eta)
        lp0, g0, st0 = ref.logp_grad(theta, eta)
        p0 = np.random.default_rng(5).standard_normal(spec.n_params).astype(np.float32)
        outs = [ref.hmc_step(1e-5, 4, p0=p0, log_u=-1e30, trace=True), ref.hmc_step(1e-5, 3)]
        th_ref = ref.get_state()
        rows = [tuple(r[name + "_rows"]) for r in ranks]
        assert rows[0][0] == 0 and rows[0][1] == rows[1][0] and rows[1][1] == n and rows[0][1] % 16 == 0, rows
        for r in ranks:
            assert int(r[name + "_allreduce_per_pass"]) == 1               # gradient row + statistic in ONE collective
            assert str(r[name + "_kernel"]) == ref.kernel_name
            assert abs(float(r[name + "_lp"]) - lp0) <= 1e-7 * abs(lp0) + 1e-6
            assert abs(float(r[name + "_st"]) - st0) <= 1e-9 * abs(st0)
            np.testing.assert_allclose(r[name + "_g"], g0, rtol=0, atol=4e-6 * np.abs(g0).max())       # same sums, another order
            np.testing.assert_allclose(r[name + "_trace"], outs[0]["trace_logp"], rtol=1e-7, atol=1e-5)
            for k in range(2):
                assert abs(r[name + "_lar"][k] - outs[k]["log_accept_ratio"]) <= 2e-3 + 1e-5 * abs(outs[k]["log_accept_ratio"])
                assert r[name + "_acc"][k] == outs[k]["accepted"]
            np.testing.assert_allclose(r[name + "_theta"], th_ref, rtol=0, atol=2e-6 * max(1.0, np.abs(th_ref).max()))
        # both ranks walked the SAME chain, bit for bit (identical reduced operands on every rank)
        np.testing.assert_array_equal(ranks[0][name + "_theta"], ranks[1][name + "_theta"])
        np.testing.assert_array_equal(ranks[0][name + "_g"], ranks[1][name + "_g"])
        if spec.n_hypers:
            h = ref.hyper_step(1e-5, 5, p0=np.random.default_rng(6).standard_normal(spec.n_hypers).astype(np.float32), log_u=-1e30)
            o3 = ref.hmc_step(1e-5, 3, p0=p0, log_u=-1e30, trace=True)
            th3 = ref.get_state()
            for r in ranks:
                assert abs(float(r[name + "_hlar"]) - h["log_accept_ratio"]) <= 2e-3 + 1e-5 * abs(h["log_accept_ratio"])
                np.testing.assert_allclose(r[name + "_eta"], ref.get_hypers(), rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(r[name + "_trace3"], o3["trace_logp"], rtol=2e-7, atol=1e-4)
                np.testing.assert_allclose(r[name + "_theta3"], th3, rtol=0, atol=4e-6 * max(1.0, np.abs(th3).max()))
        # and the chain they walked is the oracle's: value and gradient of the whole data set
        lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)
        assert abs(float(ranks[0][name + "_lp"]) - lp64) <= 4e-6 * abs(lp64) + 1e-3
        np.testing.assert_allclose(ranks[0][name + "_g"], g64, rtol=0, atol=1e-4 * np.abs(g64).max())
        ref.close()
