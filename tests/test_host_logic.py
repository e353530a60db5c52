"""Host-side logic of the drop-in API, on CPU: network.train's order of
operations / file format against the oracle's restatement (with a test double
in place of the native chain), dual averaging, and the multi-process sample
gather over gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import tbnn_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeChain:
    """test double for _native.Chain: deterministic 'transitions' (state += 1)"""

    def __init__(self, P, H):
        self.P, self.H, self.n = P, H, 0
        self.theta = np.zeros(P, np.float32)
        self.eta = np.zeros(H, np.float32)
        self.calls = []

    def set_data(self, X, Y): pass
    def set_validation(self, X, Y): pass       # nv stays 0: train() takes the host metrics path
    def set_state(self, t): self.theta = self.base = np.array(t, np.float32); self.k = 0
    def set_hypers(self, e): self.eta = self.ebase = np.array(e, np.float32); self.ke = 0
    def get_state(self): return self.theta.copy()
    def get_hypers(self): return self.eta.copy()

    def hmc_step(self, eps, L, **kw):
        self.calls.append(("w", eps, L))
        self.k += 1
        self.theta = (self.base + self.k).astype(np.float32)
        return dict(accepted=1, log_accept_ratio=-0.1, accept_prob=float(np.exp(-0.1)), sjd=float(self.P))

    def hyper_step(self, eps, L, **kw):
        self.calls.append(("h", eps, L))
        self.ke += 1
        self.eta = (self.ebase + self.ke).astype(np.float32)
        return dict(accepted=1, log_accept_ratio=-0.3, accept_prob=float(np.exp(-0.3)))

    def forward(self, X, theta=None):
        return np.zeros((1, len(X)), np.float32)

    def close(self): pass


def build_net():
    from tensorbnn_amd.network import network
    from tensorbnn_amd.layer import DenseLayer
    from tensorbnn_amd.activationFunctions import Relu
    X = np.linspace(-1, 1, 8).reshape(8, 1)
    net = network(np.float32, 1, X, np.sin(X), X, np.sin(X))
    net.add(DenseLayer(1, 10, seed=1000))
    net.add(Relu())
    net.add(DenseLayer(10, 10, seed=2000))
    net.add(Relu())
    net.add(DenseLayer(10, 1, seed=3000))
    return net


def test_add_builds_reference_state_layout():
    net = build_net()
    assert [s.shape for s in net.states] == [(10, 1), (10, 1), (10, 10), (10, 1), (1, 10), (1, 1)]
    assert len(net.hyperStates) == 12 and all(h.shape == (1,) for h in net.hyperStates)
    assert net._dense == [[1, 10, 1, 0], [10, 10, 1, 0], [10, 1, 0, 0]]
    assert [l.name for l in net.layers] == ["dense", "relu", "dense", "relu", "dense"]


def test_unsupported_plugins_fail_loudly():
    from tensorbnn_amd.activationFunctions import Softmax as Elu, Relu
    from tensorbnn_amd.network import network
    with pytest.raises(NotImplementedError):
        Elu()
    X = np.zeros((2, 1))
    net = network(np.float32, 1, X, X, X, X)
    with pytest.raises(NotImplementedError):
        net.add(Relu())          # activation with no dense layer to fuse into
    with pytest.raises(TypeError):
        network(np.float64, 1, X, X, X, X)


def test_train_loop_order_and_file_format(tmp_path, monkeypatch):
    from tensorbnn_amd.likelihood import GaussianLikelihood
    net = build_net()
    fake = FakeChain(141, 13)
    monkeypatch.setattr(type(net), "_ensure_chain", lambda self, likelihood=None: fake)
    net.setupMCMC(stepSizeStart=1e-3, stepSizeMin=1e-4, stepSizeMax=1e-2, stepSizeOptions=10, leapfrogStart=20,
                  leapfogMin=10, leapFrogMax=30, leapfrogIncrement=1, hyperStepSize=0.01, hyperLeapfrog=7, burnin=2,
                  averagingSteps=2)
    monkeypatch.chdir(tmp_path)
    theta0 = net._theta().copy()
    rec = net.train(11, 2, GaussianLikelihood(sd=0.1), folderName="run", networksPerFile=2, displaySkip=100, verbose=False)
    assert len(rec) == 11
    # order per epoch: weight transition then hyper transition (network.py:570-582)
    assert [c[0] for c in fake.calls[:4]] == ["w", "h", "w", "h"]
    assert fake.calls[1][2] == 7 and fake.calls[0][2] == 20
    # reference-format folder, read back with the predictor.py:43-113 algorithm
    mats, hyp = o.load_networks(str(tmp_path / "run"))
    assert open(tmp_path / "run" / "architecture.txt").read().split() == ["dense", "relu", "dense", "relu", "dense"]
    assert mats[0].shape[0] == 4 and len(hyp) == 4          # iters 4,6,8,10 (12 would be in the invisible last file)
    for k, it in enumerate((4, 6, 8, 10)):
        got = np.concatenate([m[k].reshape(-1) for m in mats])
        np.testing.assert_allclose(got, theta0 + it, rtol=1e-6)
        assert hyp[k].shape == (13,)
    # byte-identical to the oracle's restatement of the writer
    w = o.SampleWriter(str(tmp_path / "ora"), [s.shape for s in net.states], [l.name for l in net.layers], 13, 2, 2, 2)
    eta0 = np.concatenate([np.full(1, v, np.float32) for v in ([0, 0.5 ** 0.5, 0, 0.5 ** 0.5] * 3 + [0.1 ** 0.5])])
    for it in range(1, 12):
        th = theta0 + it
        sts, off = [], 0
        for s in net.states:
            sts.append(th[off:off + s.size].reshape(s.shape)); off += s.size
        w.after_epoch(it, sts, [eta0[i:i + 1] + it for i in range(13)])
    w.close()
    for f in sorted(os.listdir(tmp_path / "ora")):
        assert open(tmp_path / "ora" / f, "rb").read() == open(tmp_path / "run" / f, "rb").read(), f


def test_add_after_train_keeps_eta_layout(tmp_path, monkeypatch):
    """a layer added after a first train() must not end up behind the likelihood's hypers -- eta is
    (layer rows ..., likelihood hypers) whatever the call order"""
    from tensorbnn_amd.layer import DenseLayer
    from tensorbnn_amd.likelihood import GaussianLikelihood
    net = build_net()
    seen = []

    def chain_for(self, likelihood=None):
        H = sum(h.size for h in self.hyperStates)
        c = FakeChain(sum(s.size for s in self.states), H)
        seen.append(c)
        return c

    monkeypatch.setattr(type(net), "_ensure_chain", chain_for)
    net.setupMCMC(stepSizeStart=1e-3, leapfrogStart=5, leapfogMin=5, leapFrogMax=6, burnin=2, averagingSteps=2)
    monkeypatch.chdir(tmp_path)
    net.train(1, 1, GaussianLikelihood(sd=0.1), adjustHypers=False, verbose=False)
    assert seen[-1].H == 13 and seen[-1].ebase[-1] == pytest.approx(0.1 ** 0.5)
    net._dense[-1][2] = 1                                   # (test double: pretend a Relu followed the last layer)
    lay = DenseLayer(1, 1, seed=4000)
    lay.hypers = np.asarray(lay.hypers, np.float32) + 7.0   # recognisable rows
    net.add(lay)
    net.train(1, 1, GaussianLikelihood(sd=0.1), adjustHypers=False, verbose=False)
    eta = seen[-1].ebase
    assert eta.size == 17 and eta[-1] == pytest.approx(0.1 ** 0.5)                 # likelihood hyper last
    np.testing.assert_allclose(eta[12:16], np.asarray(lay.hypers, np.float32).reshape(-1))    # the new layer's rows before it


def test_dual_averaging_matches_oracle():
    net = build_net()
    net.setupMCMC(hyperStepSize=0.01, burnin=100)
    st = o.DualAveragingState(hyper_step_size=0.01, burnin=100)
    rng = np.random.default_rng(0)
    for ep in range(30):
        lar = float(rng.normal(-0.5, 1.0))
        a1 = net._dual_averaging(ep, lar)
        a2 = o.dual_averaging_update(st, ep, lar)
        assert abs(a1 - a2) < 1e-6
        assert abs(float(net.hyper_step_size) - st.eps_h) <= 1e-5 * st.eps_h
        assert abs(float(net.h) - st.h) < 1e-6


GLOO_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tensorbnn_amd import parallel
rank, world, _ = parallel.init_distributed("gloo")
class Fake:
    def __init__(s, r): s.r = r; s.k = 0
    def get_state(s): return np.full(6, 10 * s.r + s.k, np.float32)
    def get_hypers(s): return np.full(2, 100 * s.r + s.k, np.float32)
g = parallel.SampleGatherer(6, 2, device="cpu")
ch = Fake(rank)
for k in range(3):
    ch.k = k
    g(ch, k)
st = g.stacked()
assert st.shape == (3, world, 8), st.shape
for k in range(3):
    for c in range(world):
        assert np.all(st[k, c, :6] == 10 * c + k) and np.all(st[k, c, 6:] == 100 * c + k)
if rank == 0:
    parallel.write_chain_folders(sys.argv[2], st, [(2, 2), (2, 1)], ["dense"], 2)
torch.distributed.barrier()
# one marker file per rank (the launcher forwards the ranks' stdout through pipes and may interleave two lines)
os.makedirs(sys.argv[2], exist_ok=True)
open(os.path.join(sys.argv[2], f"rank{rank}.ok"), "w").write("ok")
'''


@pytest.mark.parametrize("world", [2, 8])
def test_multi_process_gloo_gather(tmp_path, world):
    """N>1 path on CPU: world_size-2 (and 8: BASELINE configs[2]'s chain count) gloo all-gather of the sampled states, chain-major,
    and the per-chain reference-format folders rank 0 writes."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    import socket
    with socket.socket() as sk:                      # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, str(tmp_path / "out")]
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all(os.path.exists(tmp_path / "out" / f"rank{c}.ok") for c in range(world)), r.stdout + r.stderr
    for c in range(world):
        mats, hyp = o.load_networks(str(tmp_path / "out" / f"chain{c}"))
        assert mats[0].shape == (3, 2, 2) and mats[1].shape == (3, 2, 1) and len(hyp) == 3
        for k in range(3):
            assert np.all(mats[0][k] == 10 * c + k) and np.all(hyp[k] == 100 * c + k)


def test_jit_family_choice_and_source():
    """host side of tensorbnn_amd/jit.py: which kernel family a shape gets and the generated translation unit"""
    from tensorbnn_amd import jit, _native as nat
    assert jit.families([5, 50, 50, 50, 1])[0] == "fast3"
    assert jit.families([5, 20, 3]) == ["fast"]                    # 3 outputs: MFMA last layer
    assert jit.families([10, 200, 200, 200, 1]) == ["wide"]
    assert jit.families([20, 100, 100, 2]) == ["mid", "wide"]      # 63 dW tiles: one wave's AccVGPRs, 134 KB of LDS
    assert jit.families([8, 80, 80, 2]) == ["mid", "wide"]
    assert jit.families([20, 111, 100, 2]) == ["mid", "wide"] and jit.families([20, 112, 112, 2]) == ["wide"]   # 7 x 8 + 14 tiles do not fit
    assert jit.families([8, 64, 64, 64, 2]) == ["mid", "wide"] and jit.families([8, 64, 64, 64, 64, 2]) == ["wide"]
    assert jit.families([40, 300, 3]) == []                        # fan-in 40, 300-wide: generic kernel
    layers = [(8, 64, nat.ACT_RELU, nat.PRIOR_CAUCHY), (64, 48, nat.ACT_RELU, nat.PRIOR_CAUCHY), (48, 2, nat.ACT_SIGMOID, nat.PRIOR_CAUCHY)]
    dims, hact, lact, bern = jit.shape_of(layers, nat.LIK_BERNOULLI)
    assert (dims, hact, lact, bern) == ([8, 64, 48, 2], nat.ACT_RELU, nat.ACT_SIGMOID, 1)
    src = jit.source(dims, hact, lact, bern, "wide")
    assert "Shape<1, 3, true, 8, 64, 48, 2>" in src and "JitWide<S>::fill" in src and "tbnn_jit_ops" in src
    assert "JitMid<S>::fill" in jit.source(dims, hact, lact, bern, "mid")
    mixed = [(4, 8, nat.ACT_RELU, 0), (8, 8, nat.ACT_TANH, 0), (8, 1, nat.ACT_NONE, 0)]
    # hidden layers with different activations (round 6): the packed per-layer code, 3 bits per hidden layer (csrc/kernels_fast.hpp: Shape::act)
    dims, hact, lact, bern = jit.shape_of(mixed, nat.LIK_GAUSSIAN)
    assert hact == jit.ACT_PACKED | nat.ACT_RELU | (nat.ACT_TANH << 3) and lact == nat.ACT_NONE
    assert f"Shape<{hact}, 0, false, 4, 8, 8, 1>" in jit.source(dims, hact, lact, bern, "fast3")
    same = [(4, 8, nat.ACT_TANH, 0), (8, 8, nat.ACT_TANH, 0), (8, 1, nat.ACT_NONE, 0)]
    assert jit.shape_of(same, nat.LIK_GAUSSIAN)[1] == nat.ACT_TANH          # one activation: the plain code, the kernels of every round before
    deep = [(4, 8, nat.ACT_RELU if i % 2 else nat.ACT_TANH, 0) for i in range(1)] + [(8, 8, nat.ACT_RELU if i % 2 else nat.ACT_TANH, 0) for i in range(9)] + [(8, 1, 0, 0)]
    assert jit.shape_of(deep, nat.LIK_GAUSSIAN) is None                     # 10 hidden layers with mixed activations: the layered family


def test_predictor_reweight_without_likelihood(tmp_path):
    """predictor.reweight / trainProbs, the path the reference can run (likelihood=None, predictor.py:157-273):
    weights = exp(sum_layers hyperprobs_new - hyperprobs_train), normalised -- checked against a direct evaluation.
    Layers from customLayerDict (user plug-ins, predictor.py:30-36) are judged by their own Python calculateHyperProbs: that
    is the route this host test walks; the built-in dense layers go through tbnn_hyper_probs_many on the device
    (tests/test_gpu_metrics.py::test_reweight_on_device)."""
    from tensorbnn_amd.predictor import predictor
    from tensorbnn_amd.layer import CauchyDenseLayer, GaussianDenseLayer

    class UserCauchy(CauchyDenseLayer):
        pass

    class UserGaussian(GaussianDenseLayer):
        pass
    rng = np.random.default_rng(11)
    shapes = [(4, 3), (4, 1), (2, 4), (2, 1)]
    names = ["dense", "relu", "dense"]
    w = o.SampleWriter(str(tmp_path / "run"), shapes, names, 9, 1, 1, 2)
    nets = []
    for it in range(1, 7):
        sts = [(0.5 * rng.standard_normal(sh)).astype(np.float32) for sh in shapes]
        # near the Gaussian hyper-priors' modes: the un-normalised exp() of the reference (predictor.py:267) stays finite
        hyp = [np.float32(v + 0.02 * rng.standard_normal(1)) for v in ([0.02, 1.0, -0.02, 0.99] * 2 + [0.3])]
        nets.append((sts, hyp))
        w.after_epoch(it, sts, hyp)
    w.close()
    p = predictor(str(tmp_path / "run") + "/", customLayerDict={"dense": UserCauchy, "denseGaussian": UserGaussian})
    assert p.numNetworks >= 4 and p.extractParameters()[0].shape[1:] == (4, 3)
    (tmp_path / "arch2.txt").write_text("denseGaussian\nrelu\ndenseGaussian\n")
    wts = p.reweight(str(tmp_path / "arch2.txt"), n=1, likelihood=None)
    assert wts.shape == (p.numNetworks,) and abs(wts.sum() - 1) < 1e-5 and np.all(wts >= 0)
    # direct evaluation
    lw = []
    for k in range(p.numNetworks):
        t = [m[k] for m in p.matrices]
        h = p.hypers[k]
        old = sum(float(CauchyDenseLayer(1, 1).calculateHyperProbs(h[4 * j:4 * j + 4], t[2 * j:2 * j + 2])) for j in range(2))
        new = sum(float(GaussianDenseLayer(1, 1).calculateHyperProbs(h[4 * j:4 * j + 4], t[2 * j:2 * j + 2])) for j in range(2))
        lw.append(new - old)
    ref = np.exp(np.array(lw) - max(lw)); ref /= ref.sum()
    np.testing.assert_allclose(wts, ref, rtol=2e-4, atol=1e-7)
    # the architecture is restored afterwards (predictor.py:271)
    assert [l.name for l in p.layers] == names
    assert p._chain is None                       # no native chain was needed for user-defined layers


def test_autocorr_restatement():
    """function_1d / integrated_time (emcee.autocorr, used at predictor.py:275-312) on an AR(1) series:
    acf(k) = rho^k, tau = (1 + rho) / (1 - rho)"""
    from tensorbnn_amd.predictor import function_1d, integrated_time
    rng = np.random.default_rng(0)
    rho, n = 0.8, 200000
    e = rng.standard_normal(n)
    x = np.empty(n); x[0] = e[0]
    for i in range(1, n):
        x[i] = rho * x[i - 1] + e[i]
    f = function_1d(x)
    assert f[0] == 1.0 and np.allclose(f[1:6], rho ** np.arange(1, 6), atol=0.02)
    tau = integrated_time(x, tol=5, quiet=True)
    assert tau.shape == (1,) and abs(tau[0] - (1 + rho) / (1 - rho)) < 0.6
    with pytest.raises(ValueError):
        integrated_time(x[:50], tol=50)


def test_adapter_grid_search_threads_agree(monkeypatch):
    """the threaded UCB grid scan (adapter.cpp gridSearch, paramAdapter.py:158-196) returns the sequential scan's (eps, L):
    a grid large enough to be split (100 x 496 points x history), one thread against the default thread count"""
    from tensorbnn_amd.paramAdapter import paramAdapter
    def run(threads):
        if threads is None: monkeypatch.delenv("TBNN_ADAPTER_THREADS", raising=False)
        else: monkeypatch.setenv("TBNN_ADAPTER_THREADS", str(threads))
        ad = paramAdapter(1e-3, 1000, 1e-4, 1e-2, 100, 100, 10000, 20, 2, 50, a=4, delta=0.1, randomSteps=3, seed=5)
        rng = np.random.default_rng(3)
        st = rng.standard_normal(300).astype(np.float32)
        out = []
        for i in range(80):
            st = st + (0.02 + 0.01 * np.sin(i)) * rng.standard_normal(300).astype(np.float32)
            out.append(tuple(map(float, ad.update(st))))
        return out
    a, b, c = run(1), run(None), run(3)
    assert a == b == c
    assert len(set(a[10:])) > 3          # the search did move (eps, L) around


def test_burned_fixtures_are_consistent():
    """the committed burned-in chain states bench.py and the free-running parity tests start from (tools/make_burned.py):
    shapes match the workloads, the step size is one the recorded scan (or tools/epscal.py's note) supports"""
    import json
    from tensorbnn_amd.workloads import WORKLOADS, burned_state
    gold = os.path.join(ROOT, "tests", "golden")
    for cfg, wl in WORKLOADS.items():
        b = burned_state(cfg, os.path.join(ROOT, "tests", "golden"))
        if cfg in ("w300", "mc10", "wm10", "oh100", "wf50"):   # profile-only shapes (tools/profile_round.sh): run from the initial state
            assert b is None
            continue
        assert b is not None, cfg
        dims = wl["dims"]
        P = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
        H = 4 * (len(dims) - 1) + (1 if wl["lik"] == 0 else 0)
        assert b["theta"].shape == (P,) and b["eta"].shape == (H,) and b["theta"].dtype == np.float32
        assert np.all(np.isfinite(b["theta"])) and np.all(np.isfinite(b["eta"]))
        assert int(b["L"]) == wl["L"] and int(b["epochs"]) >= 100
        scan = b["scan"]
        assert scan.shape[1] == 2 and scan[:, 0].min() * 0.5 <= float(b["eps"]) <= scan[:, 0].max()
        assert np.any((scan[:, 1] >= 0.6) & (scan[:, 1] <= 0.9))            # the scan found the regime SURVEY 8(d) asks for
        meta = json.load(open(os.path.join(gold, f"{cfg}_burned.json")))
        assert meta["rows"] == wl["n"] and meta["dims"] == dims and abs(meta["eps"] - float(b["eps"])) < 1e-12
        if wl["hyper"]:
            assert "da_step" in b and float(b["da_step"]) > 0


def test_bench_host_helpers():
    """bench.py's host-side pieces that need no GPU: algorithmic FLOP per leapfrog step, CPU topology"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_flops([5, 50, 50, 50, 1], 100_000) == pytest.approx(3.13e9, rel=2e-3)
    assert bench.algorithmic_flops([10, 200, 200, 200, 1], 1_000_000) == pytest.approx(4.892e11, rel=1e-3)
    assert bench.algorithmic_flops([20, 100, 100, 2], 500_000) == pytest.approx(3.46e10, rel=2e-3)
    logical, physical = bench.host_cpus()
    assert 1 <= physical <= logical
    assert set(bench.CONFIG_KEY) >= {"c1", "c2", "c4", "c5"}


def test_bench_quotes_committed_profiles_only_for_the_loaded_build(tmp_path):
    """roofline.frac_rocprof / roofline.traffic come from committed rocprofv3 summaries; they are quoted only when those were
    measured on the build that is loaded (tbnn_build_id written by tools/rocprof_summary.py), and are null otherwise -- a
    kernel changed without re-profiling must not carry the old kernel's numbers"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from tensorbnn_amd import _native as nat, build as b
    bid = nat.build_id()
    assert len(bid) == 16 and bid == b.sources_id(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value",
                                                     "-Wno-unused-result"])        # the library reports the sources it was built from
    ku = {"c2": {"us": 44.3, "kernels": {"k_fwd_bwd_fast3": 44.3}, "source": "profiles/x_c2_kernel_stats.csv"},
          "c5": {"us": 362.0, "kernels": {"k_fwd_bwd_mid": 362.0}, "source": "profiles/x_c5_kernel_stats.csv"},
          "_build": {"build_id": bid, "lib_sha256": "0" * 64, "git_head": "abc", "tag": "x"}}
    tr = {"c2": {"hbm_bytes_per_launch": 12593417}, "_build": dict(ku["_build"])}
    json.dump(ku, open(tmp_path / "rocprof_kernel_us.json", "w")); json.dump(tr, open(tmp_path / "pmc_traffic.json", "w"))
    got = bench.committed_profile("c2", bid, str(tmp_path))
    assert got["match"] and got["kernel_us"] == 44.3 and got["traffic"] == 12593417 and got["source"].endswith("x_c2_kernel_stats.csv")
    assert bench.committed_profile("c5g", bid, str(tmp_path))["kernel_us"] == 362.0           # configs[4]'s kernels on other priors
    stale = bench.committed_profile("c2", "f" * 16, str(tmp_path))                           # another build is loaded
    assert not stale["match"] and stale["kernel_us"] is None and stale["traffic"] is None and stale["profiled_build"] == bid
    tr["_build"]["build_id"] = "e" * 16                                                      # PMC pass from an older build than the trace
    json.dump(tr, open(tmp_path / "pmc_traffic.json", "w"))
    half = bench.committed_profile("c2", bid, str(tmp_path))
    assert half["kernel_us"] == 44.3 and half["traffic"] is None
    del ku["_build"]                                                                         # a file from before round 4: never quoted
    json.dump(ku, open(tmp_path / "rocprof_kernel_us.json", "w"))
    assert bench.committed_profile("c2", bid, str(tmp_path))["kernel_us"] is None
    assert bench.committed_profile("c2", bid, str(tmp_path / "nowhere"))["kernel_us"] is None
    # the committed files under profiles/ carry a build id (whatever it is)
    for f in ("rocprof_kernel_us.json", "pmc_traffic.json"):
        assert "_build" in json.load(open(os.path.join(ROOT, "profiles", f))), f


def test_bench_line_fits_driver_tail():
    """the stdout line of bench.py keeps every config's value / roofline fraction / CPU baseline within 2 KB (the driver's
    record keeps a 2-KB tail); checked on the committed full record of round 2 with the rocprof fields filled in"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r02b_bench_full.json")))
    for r in [full] + list(full["secondary"].values()):
        if r.get("roofline") and r["roofline"].get("frac") is not None:
            r["roofline"].update(frac_rocprof=0.4291, rocprof_kernel_us=4849.123, rocprof_source="profiles/r03z_c4_kernel_stats.csv",
                                 profile_build_match=True)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= 2300
    tail = text[-2048:]
    for key in ("configs[3]", "configs[4]", "configs[4]g", "configs[0]"):
        assert f'"{key}": {{"value"' in tail
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["roofline"]["frac_rocprof"] == 0.4291 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    # the round-4 record (one more secondary entry: 64 chains on one GPU): the WHOLE line inside the 2-KB tail, so that it parses
    full4 = json.load(open(os.path.join(ROOT, "profiles", "r04i_bench_full.json")))
    text4 = json.dumps(bench.compact_line(full4))
    assert len(text4) <= 2048 and '"configs[0]x64": {"value"' in text4 and json.loads(text4)["roofline"]["profile_build_match"] is True


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it starts torch.distributed.run --nproc-per-node N as a child
    (never exec) with the same arguments"""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # (conftest caps OpenBLAS for the test process itself)
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # thread caps for N ranks on a CPU-limited cgroup
        assert 1 <= int(seen["env"][var]) <= 4
    # more ranks than GPUs: the parent refuses before anything is started (sysfs says how many there are; no HIP call)
    seen.clear()
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    monkeypatch.delenv("TBNN_BENCH_SINGLE_GPU", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 1 GPU(s) visible" in str(e.value.code) and not seen
    monkeypatch.setenv("TBNN_BENCH_SINGLE_GPU", "1")                                # the one-GPU rehearsal hook still spawns
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and seen


def test_visible_gpus(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024, 1024)):                                # node 0: the CPU
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\n")
    real_listdir, real_open = os.listdir, open
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(nodes)) if p == "/sys/class/kfd/kfd/topology/nodes" else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace("/sys/class/kfd/kfd/topology/nodes", str(nodes)), *a, **k))
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench._kfd_gpu_nodes() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench._kfd_gpu_nodes() == 2
    # the count that decides comes from a CHILD asking the native library (this process never touches HIP): no device here
    import builtins as _b
    monkeypatch.setattr(_b, "open", real_open); monkeypatch.setattr(os, "listdir", real_listdir)
    assert bench.visible_gpus() in (0, None)


def test_cpu_quota_warning(monkeypatch, tmp_path):
    """_native warns once per process when a math library's thread pool is larger than the container's CPU quota (the launch thread of
    a chain was frozen for 80 ms at a time on a 256-CPU box with a 16-CPU quota: NOTES.md round 4)"""
    import builtins, warnings, types
    from tensorbnn_amd import _native as nat
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO("400000 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert nat.cpu_quota() == 4.0
    fake = types.ModuleType("threadpoolctl")
    fake.threadpool_info = lambda: [{"internal_api": "openblas", "num_threads": 64}, {"internal_api": "openmp", "num_threads": 2}]
    monkeypatch.setitem(sys.modules, "threadpoolctl", fake)
    monkeypatch.setattr(nat.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(nat, "_QUOTA_WARNED", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        nat._warn_cpu_quota_once()
        nat._warn_cpu_quota_once()                      # once per process
    assert len(w) == 1 and "openblas: 64 threads" in str(w[0].message) and "OPENBLAS_NUM_THREADS" in str(w[0].message)
    # a pool within the quota: silent
    fake.threadpool_info = lambda: [{"internal_api": "openblas", "num_threads": 4}]
    monkeypatch.setattr(nat, "_QUOTA_WARNED", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        nat._warn_cpu_quota_once()
    assert not w


def test_bench_line_says_what_the_hazard_check_did():
    """the stdout line's roofline object carries the build-time MFMA hazard check in short (units checked / pairs repaired in their listings);
    the full status string stays in gpurun_out/bench_full.json"""
    sys.path.insert(0, ROOT)
    import bench
    full = ("tbnn_api.hip: listing checked (0 repaired); disassembly clean; tbnn_mid.hip: listing checked (2 repaired: R2c:2); disassembly clean; "
            "wait states inside the asm MFMAs: on (off in tbnn_tall.hip: left to the check)")
    short = bench._short_roof({"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "hazard_check": full})
    assert short["hazard_check"] == "2 units checked, 2 pairs repaired"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(short)
    assert "hazard_check" not in bench._short_roof({"bound": "mfma"})
