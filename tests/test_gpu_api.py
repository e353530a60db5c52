"""The drop-in Python API end to end on the GPU: the reference's
trainRegression.py flow (Examples/trainRegression.py:31-111), the sample files,
the predictor, accept-ratio parity against the oracle chain."""
import os

import numpy as np
import pytest

import tbnn_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arch", ["one_hidden_100", "mixed_activations"])
def test_train_flow_on_run_time_instantiated_kernels(tmp_path, monkeypatch, native, arch):
    """the example problem (11 rows) through the Python API on architectures that reach their fused kernel through jit.py since late round 6:
    1->100->1 (one wide hidden layer: narrow kernels) and 1->10 Relu->10 Tanh->1 (hidden layers with different activations)"""
    import math
    from tensorbnn_amd.activationFunctions import Relu, Tanh
    from tensorbnn_amd.layer import GaussianDenseLayer
    from tensorbnn_amd.likelihood import FixedGaussianLikelihood
    from tensorbnn_amd.metrics import SquaredError
    from tensorbnn_amd.network import network
    from tensorbnn_amd.predictor import predictor
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("TBNN_JIT", "1")
    trainIn = np.linspace(-2, 2, num=11)
    valIn = np.linspace(-2 + 2 / 30, 2.0 - 2 / 30, num=30)
    trainOut = np.sin(trainIn * math.pi * 2) * trainIn - np.cos(trainIn * math.pi)
    valOut = np.sin(valIn * math.pi * 2) * valIn - np.cos(valIn * math.pi)
    net = network(np.float32, 1, trainIn, trainOut.T, valIn, valOut.T)
    if arch == "one_hidden_100":
        net.add(GaussianDenseLayer(1, 100, seed=1000)); net.add(Tanh())
        net.add(GaussianDenseLayer(100, 1, seed=2000))
        dims, acts, want = [1, 100, 1], [o.ACT_TANH], "fast3<tanh,"
    else:
        net.add(GaussianDenseLayer(1, 10, seed=1000)); net.add(Relu())
        net.add(GaussianDenseLayer(10, 10, seed=2000)); net.add(Tanh())
        net.add(GaussianDenseLayer(10, 1, seed=3000))
        dims, acts, want = [1, 10, 10, 1], [o.ACT_RELU, o.ACT_TANH], "fast3<relu+tanh,"
    net.setupMCMC(stepSizeStart=1e-3, stepSizeMin=1e-4, stepSizeMax=1e-2, stepSizeOptions=20, leapfrogStart=20,
                  leapfogMin=10, leapFrogMax=50, leapfrogIncrement=10, hyperStepSize=0.001, hyperLeapfrog=20,
                  burnin=10, averagingSteps=5)
    rec = net.train(31, 10, FixedGaussianLikelihood(sd=0.1), metricList=[SquaredError()], adjustHypers=True, folderName="Run",
                    networksPerFile=2, displaySkip=30)
    assert len(rec) == 31 and want in net._chain.kernel_name, net._chain.kernel_name
    assert np.mean([r["main"]["accept_prob"] for r in rec]) > 0.1
    p = predictor(str(tmp_path / "Run") + "/")
    assert p.numNetworks == 2
    preds = p.predict(valIn.reshape(-1, 1))
    spec = o.make_spec(dims, acts[0], o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN)
    for l, a in zip(spec.layers[:-1], acts):
        l.act = a
    ref = o.forward(spec, p.vectors[0], valIn.reshape(-1, 1).astype(np.float32), np.float64)
    np.testing.assert_allclose(preds[0], ref, rtol=2e-5, atol=2e-5)


def test_train_regression_script_flow(tmp_path, monkeypatch, native):
    """the literal example problem: 11 rows, 1->10->10->10->1 Tanh, GaussianDenseLayer, FixedGaussianLikelihood"""
    import math
    from tensorbnn_amd.activationFunctions import Tanh
    from tensorbnn_amd.layer import GaussianDenseLayer
    from tensorbnn_amd.likelihood import FixedGaussianLikelihood
    from tensorbnn_amd.metrics import PercentError, SquaredError
    from tensorbnn_amd.network import network
    from tensorbnn_amd.predictor import predictor
    monkeypatch.chdir(tmp_path)
    trainIn = np.linspace(-2, 2, num=11)
    valIn = np.linspace(-2 + 2 / 30, 2.0 - 2 / 30, num=30)
    trainOut = np.sin(trainIn * math.pi * 2) * trainIn - np.cos(trainIn * math.pi)
    valOut = np.sin(valIn * math.pi * 2) * valIn - np.cos(valIn * math.pi)
    net = network(np.float32, 1, trainIn, trainOut.T, valIn, valOut.T)
    seed = 1000
    net.add(GaussianDenseLayer(1, 10, seed=seed)); net.add(Tanh()); seed += 1000
    for _ in range(2):
        net.add(GaussianDenseLayer(10, 10, seed=seed)); net.add(Tanh()); seed += 1000
    net.add(GaussianDenseLayer(10, 1, seed=seed))
    net.setupMCMC(stepSizeStart=1e-3, stepSizeMin=1e-4, stepSizeMax=1e-2, stepSizeOptions=20, leapfrogStart=50,
                  leapfogMin=10, leapFrogMax=100, leapfrogIncrement=10, hyperStepSize=0.001, hyperLeapfrog=20,
                  burnin=20, averagingSteps=5)
    rec = net.train(61, 10, FixedGaussianLikelihood(sd=0.1), metricList=[SquaredError(), PercentError()],
                    adjustHypers=True, folderName="TrigRegression", networksPerFile=2, displaySkip=30)
    assert len(rec) == 61 and "<tanh" in net._chain.kernel_name
    assert all(np.isfinite(r["main"]["log_accept_ratio"]) or r["main"]["log_accept_ratio"] == -np.inf for r in rec)
    assert np.mean([r["main"]["accept_prob"] for r in rec]) > 0.2
    assert any(r["L"] != 50 or abs(r["eps"] - 1e-3) > 1e-9 for r in rec[10:])       # the adapter moved (eps, L)
    p = predictor(str(tmp_path / "TrigRegression") + "/")
    assert p.numNetworks == 4 and len(p.hypers) == 4 and p.hypers[0].shape == (16,)      # iters 30,40,50,60
    preds = p.predict(valIn.reshape(-1, 1))
    assert len(preds) == 4 and preds[0].shape == (1, 30)
    # the predictor's forward equals the oracle's forward of the saved weights
    spec = o.make_spec([1, 10, 10, 10, 1], o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN)
    ref = o.forward(spec, p.vectors[0], valIn.reshape(-1, 1).astype(np.float32), np.float64)
    np.testing.assert_allclose(preds[0], ref, rtol=2e-5, atol=2e-5)
    # re-weighting with the data term (predictor.py:157-273; see predictor.py's docstring for the reference's defects):
    # same architecture + same likelihood => uniform weights; the data term is the training log-likelihood
    from tensorbnn_amd.layer import _multivariate_log_prob
    p2 = predictor(str(tmp_path / "TrigRegression") + "/", likelihood=FixedGaussianLikelihood(sd=0.1))
    wts = p2.reweight(str(tmp_path / "TrigRegression" / "architecture.txt"), trainX=trainIn.reshape(-1, 1), trainY=trainOut,
                      n=1, likelihood=FixedGaussianLikelihood(sd=0.1))
    np.testing.assert_allclose(wts, np.full(4, 0.25), rtol=1e-5)
    f0 = o.forward(spec, p.vectors[0], trainIn.reshape(-1, 1).astype(np.float32), np.float64).T
    ll0 = float(np.sum(_multivariate_log_prob(np.full(f0.shape, 0.1, np.float32), f0.astype(np.float32),
                                              trainOut.reshape(f0.shape).astype(np.float32))))
    hp0 = sum(float(GaussianDenseLayer(1, 1).calculateHyperProbs(p.hypers[0][4 * j:4 * j + 4], [p.matrices[2 * j][0], p.matrices[2 * j + 1][0]]))
              for j in range(4))
    assert abs(float(p2.weightsTrain[0]) + ll0 + hp0) <= 2e-4 * abs(ll0 + hp0) + 1e-3
    # autocorrelation over the ensemble's predictions (predictor.py:275-312)
    ac = p.autocorrelation(valIn.reshape(-1, 1), 3)
    assert ac.shape == (3,) and abs(ac[0] - 1.0) < 1e-6
    assert np.isfinite(p.autoCorrelationLength(valIn.reshape(-1, 1), 100))


def test_accept_ratio_parity_with_oracle_chain(native):
    """accept-ratio parity +-0.02: the GPU chain and the fp32 oracle chain driven by the
    same injected momenta / uniforms over 40 epochs of a down-scaled configs[1]."""
    spec, X, Y, theta, eta = o.synth_problem([5, 50, 50, 50, 1], 512)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    rng = np.random.default_rng(3)
    th = theta.copy()
    eps, L = 1.2e-4, 10
    acc_g, acc_o, dec_same = [], [], 0
    for ep in range(40):
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        lu = float(np.log(rng.random()))
        ch.set_state(th)                              # same start each epoch: compares the transition itself
        out = ch.hmc_step(eps, L, p0=p0, log_u=lu)
        ref = o.weight_step(spec, th, eta, X, Y, eps, L, p0, lu, np.float32)
        acc_g.append(out["accept_prob"]); acc_o.append(ref.accept_prob)
        dec_same += int(bool(out["accepted"]) == ref.accepted)
        th = ref.theta
    assert abs(np.mean(acc_g) - np.mean(acc_o)) <= 0.02, (np.mean(acc_g), np.mean(acc_o))
    assert 0.3 < np.mean(acc_o) < 0.999          # a non-trivial regime
    assert dec_same >= 38
    ch.close()


@pytest.mark.parametrize("case", ["gaussian_lik", "bernoulli", "fixed_gaussian_prior"])
def test_hyper_transition_changes_weight_target(native, case):
    """after an accepted hyper transition the cached (log-prob, gradient) of the current state is refreshed WITHOUT another
    pass over the rows (k_refresh_grad_after_hyper: the data-term gradient rescales with the likelihood's sigma, the prior
    terms are recomputed): the next transition's whole log-prob trace -- its first step moves along the refreshed gradient --
    must be the oracle's at the new eta; several (weight, hyper) rounds, rejects in between"""
    dims, n, act, prior, lik = {"gaussian_lik": ([1, 10, 10, 1], 200, o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_GAUSSIAN),
                                "bernoulli": ([20, 32, 16, 48, 2], 300, o.ACT_SIGMOID, o.PRIOR_CAUCHY, o.LIK_BERNOULLI),
                                "fixed_gaussian_prior": ([1, 10, 10, 10, 1], 64, o.ACT_TANH, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN)}[case]
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd)
    ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
    rng = np.random.default_rng(12)
    eps = 2e-3 if lik == o.LIK_FIXED_GAUSSIAN else 1e-4
    ch.hmc_step(eps, 3)
    for rnd in range(3):
        h = ch.hyper_step(1e-5, 5, log_u=-1e30)                 # forced accept: eta moves
        assert h["accepted"] == 1
        eta2, th2 = ch.get_hypers(), ch.get_state()
        assert np.abs(eta2 - eta).max() > 0
        p0 = (0.1 * rng.standard_normal(ch.P)).astype(np.float32)
        lu = 1e30 if rnd == 1 else float(np.log(0.7))            # a forced reject in the middle round
        out = ch.hmc_step(eps, 3, p0=p0, log_u=lu, trace=True)
        ref = o.weight_step(spec, th2, eta2, X, Y, eps, 3, p0, lu, np.float64)
        assert abs(out["logp_old"] - ref.logp_old) <= 4e-6 * abs(ref.logp_old) + 1e-3, (case, rnd)
        np.testing.assert_allclose(out["trace_logp"], ref.trace_logp, rtol=4e-6, atol=2e-3, err_msg=f"{case} round {rnd}")
        assert abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 1e-6 * abs(ref.logp_old)
        assert bool(out["accepted"]) == ref.accepted
    ch.close()


def test_bench_multi_rank_control_flow(tmp_path):
    """bench.py's N > 1 path (barrier, per-rank chains, sample gather, max-over-ranks timing) with two ranks
    sharing the one GPU of the test box (TBNN_BENCH_SINGLE_GPU=1: collectives over gloo); the real runs use
    one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # plain `bench.py --gpus 2`, as the driver runs it: bench.py starts its own two ranks (torch.distributed.run as a child)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--sampling-step", "3"]
    # torch's collectives over gloo (two ranks on one GPU); the checkpoint gather is the native tbnn_gather_samples, its
    # collective library pointed at the test stub (RCCL itself refuses two ranks on one device)
    from test_gpu_multirank import build_stub
    env = _bench_env(TBNN_BENCH_SINGLE_GPU="1", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["chains"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["sample_gather"].startswith("tbnn_gather_samples") and "secondary" not in d
    assert d["value"] > 0 and d["roofline"]["frac"] > 0.05
    assert d["ranks"]["nccl_comm_count"] == [2] and d["ranks"]["native_gather"] is True
    assert 0 < d["ranks"]["steps_per_s_min"] <= d["ranks"]["steps_per_s_max"]


def _bench_env(**extra):
    from conftest import wait_gpu_quiet
    from test_gpu_multirank import build_stub
    wait_gpu_quiet()
    env = dict(os.environ, TBNN_RCCL_LIB=build_stub(), **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS"):
        env.pop(k, None)
    return env


def test_bench_three_ranks_rehearsal(tmp_path):
    """The driver's multi-GPU invocation rehearsed on the one GPU of the test box: `bench.py --gpus 3` starts its own three ranks
    (torch.distributed.run as a child), the unique id travels over torch.distributed, every rank joins the native communicator
    (stand-in collective library: RCCL refuses two ranks on one device), gathers through tbnn_gather_samples inside the timed
    region, and rank 0 prints ONE line that says what every rank saw.  Three, not eight: the GPU box admits at most six processes
    on its card, and this test's own process, the launcher (importing torch opens the device) and the ranks all count; nothing in
    the path depends on the rank count beyond the collective library's own limit (tests/stubccl: 8)."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "6", "--warmup", "2", "--sampling-step", "2"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_bench_env(TBNN_BENCH_SINGLE_GPU="1"), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["chains"] == 3 and d["scaling"] == "weak" and d["steps"] == 6
    assert d["config"]["sample_gather"].startswith("tbnn_gather_samples")
    assert d["ranks"]["nccl_comm_count"] == [3] and d["ranks"]["native_gather"] is True
    assert 0 < d["ranks"]["steps_per_s_min"] <= d["ranks"]["steps_per_s_max"]
    # value = leapfrog steps of ALL ranks over the slowest rank's wall time
    assert abs(d["value"] - 3 * 6 * 50 / (d["ms_per_step"] * 6 / 1e3)) <= 1e-3 * d["value"]
    full = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")][-1]
    assert len(full["ranks"]["steps_per_s"]) == 3 and full["ranks"]["nccl_comm_count"] == [3, 3, 3]
    assert full["ranks"]["omp_num_threads"] is not None            # thread caps reached the ranks
    print(f"3-rank rehearsal: {time.time() - t0:.0f} s, per-rank steps/s {full['ranks']['steps_per_s']}")


def test_bench_distributed_path_on_the_real_rccl(tmp_path):
    """The N > 1 code path of bench.py against the REAL collective library, as far as one GPU allows: world = 1 with
    TBNN_BENCH_FORCE_DIST=1 -- torch's "nccl" process group (RCCL) on the device, the unique id through broadcast_object_list,
    ncclCommInitRank / ncclAllGather / ncclCommCount through the C ABI's own plumbing inside the timed region, the per-rank
    diagnostics gathered over the process group.  (More than one rank needs more than one GPU: RCCL refuses two ranks per device.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bench_env(TBNN_BENCH_FORCE_DIST="1")
    env.pop("TBNN_RCCL_LIB", None)                                   # the real librccl.so (torch's copy is re-used)
    env.pop("TBNN_BENCH_SINGLE_GPU", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--sampling-step", "2",
           "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["sample_gather"].startswith("tbnn_gather_samples (RCCL")
    assert d["ranks"]["nccl_comm_count"] == [1] and d["ranks"]["native_gather"] is True and d["value"] > 0
    full = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")][-1]
    assert full["ranks"]["collective_library"] == "librccl (RCCL)"


def test_bench_rank_death_is_a_nonzero_exit(tmp_path):
    """a rank that dies inside the timed region (test hook TBNN_BENCH_FAIL_RANK) takes the whole job down: non-zero exit code,
    no result line, no hang"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--sampling-step", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_bench_env(TBNN_BENCH_SINGLE_GPU="1", TBNN_BENCH_FAIL_RANK="1"), cwd=root)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_more_ranks_than_gpus_fails_fast():
    """`bench.py --gpus 8` on a box with fewer GPUs: refused by the parent before any rank is started (one rank per GPU; it never
    doubles ranks up on a device by itself), non-zero, within seconds"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = _bench_env()
    env.pop("TBNN_BENCH_SINGLE_GPU", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout), r.stderr[-500:]
    assert time.time() - t0 < 120          # (seconds of work; the bound leaves room for a first `import torch` on a cold box)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_classification_flow(tmp_path, monkeypatch, native):
    """the classification use (docs/ClassificationExample.md:103-173 with the activations this build supports): Relu hidden
    layers, Sigmoid output, BernoulliLikelihood (no hyper of its own), Accuracy metric, samples on disk, predictor"""
    from tensorbnn_amd.activationFunctions import Relu, Sigmoid
    from tensorbnn_amd.layer import DenseLayer
    from tensorbnn_amd.likelihood import BernoulliLikelihood
    from tensorbnn_amd.metrics import Accuracy
    from tensorbnn_amd.network import network
    from tensorbnn_amd.predictor import predictor
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(7)
    X = rng.standard_normal((600, 4)).astype(np.float32)
    lab = (X[:, 0] + 0.5 * X[:, 1] * X[:, 2] > 0).astype(np.float32)
    Y = np.stack([lab, 1 - lab], axis=1)                      # two outputs, one-hot
    net = network(np.float32, 4, X[:500], Y[:500], X[500:], Y[500:])
    net.add(DenseLayer(4, 12, seed=1000)); net.add(Relu())
    net.add(DenseLayer(12, 12, seed=2000)); net.add(Relu())
    net.add(DenseLayer(12, 2, seed=3000)); net.add(Sigmoid())
    net.setupMCMC(stepSizeStart=2e-3, stepSizeMin=5e-4, stepSizeMax=5e-3, stepSizeOptions=10, leapfrogStart=20, leapfogMin=10,
                  leapFrogMax=40, leapfrogIncrement=5, hyperStepSize=1e-4, hyperLeapfrog=10, burnin=20, averagingSteps=5)
    acc = Accuracy()
    rec = net.train(60, 5, BernoulliLikelihood(), metricList=[acc], adjustHypers=True, folderName="cls", networksPerFile=2,
                    displaySkip=30, verbose=True)           # files rotate every 10 epochs after burn-in (network.py:609-646)
    assert len(rec) == 60 and net._chain.H == 12                # 3 dense layers x 4 hypers, none from the likelihood
    assert np.mean([r["main"]["accept_prob"] for r in rec]) > 0.3
    p = predictor(str(tmp_path / "cls") + "/", likelihood=BernoulliLikelihood())
    assert p.numNetworks == 6 and p.hypers[0].shape == (12,)      # the summary of the rotation at iter 51: iters 25..50
    preds = np.array(p.predict(X[500:]))                       # [networks, 2, rows]
    assert preds.shape[1:] == (2, 100) and np.all((preds >= 0) & (preds <= 1))
    # the posterior-mean classifier beats chance on the held-out rows
    mean_p = preds.mean(axis=0)
    assert np.mean((mean_p[0] > mean_p[1]) == (lab[500:] > 0.5)) > 0.7
    # same forward as the oracle for a saved network
    spec = o.make_spec([4, 12, 12, 2], o.ACT_RELU, o.PRIOR_CAUCHY, o.LIK_BERNOULLI, final_act=o.ACT_SIGMOID)
    np.testing.assert_allclose(preds[0], o.forward(spec, p.vectors[0], X[500:], np.float64), rtol=2e-5, atol=2e-5)


def _free_device_bytes():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_create_destroy_returns_device_memory(native):
    """tbnn_destroy frees what tbnn_create / tbnn_set_data / the first transition allocated, for every kernel family and for a chain
    group: five create-run-destroy cycles leave the device's free memory where one cycle left it"""
    cases = [([5, 50, 50, 50, 1], 20000, o.LIK_GAUSSIAN, "fast3<"), ([20, 100, 100, 2], 8000, o.LIK_BERNOULLI, "mid<"),
             ([784, 20, 20, 1], 4000, o.LIK_BERNOULLI, "tall<"), ([10, 200, 200, 200, 1], 20000, o.LIK_GAUSSIAN, "wide<"),
             ([8, 300, 300, 1], 5000, o.LIK_GAUSSIAN, "layered<"), ([3, 7, 2], 900, o.LIK_GAUSSIAN, None)]
    probs = [(o.synth_problem(d, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik), fam) for d, n, lik, fam in cases]

    def cycle():
        for (spec, X, Y, theta, eta), fam in probs:
            layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
            ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=False)
            if fam:
                assert fam in ch.kernel_name, ch.kernel_name
            ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
            ch.hmc_step(1e-6, 2, trace=True)
            ch.hyper_step(1e-4, 2)
            ch.set_validation(X[:64], Y[:64]); ch.predict(1)
            ch.close()
        spec, X, Y, theta, eta = probs[0][0]
        layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
        grp = native.ChainGroup(layers, 4, likelihood=spec.likelihood)
        grp.set_data(X, Y); grp.set_state(np.tile(theta, (4, 1))); grp.set_hypers(np.tile(eta, (4, 1)))
        grp.hmc_run(1e-6, 2, 2)
        grp.close()

    cycle()                                    # code objects, pinned staging, the runtime's own pools
    base = _free_device_bytes()
    for _ in range(5):
        cycle()
    lost = base - _free_device_bytes()
    assert lost <= 32 << 20, f"{lost / 2**20:.1f} MiB of device memory did not come back"


@pytest.mark.parametrize("dims", [[5, 50, 50, 50, 1], [1, 10, 10, 1]])
def test_fused_burst_timing_leaves_the_chain_untouched(native, dims):
    """tbnn_debug_fused_burst (bench.py's throughput figure): a positive time per pass, and the chain goes on exactly as one that was never timed"""
    spec, X, Y, theta, eta = o.synth_problem(dims, 2048)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    recs = []
    for timed in (False, True):
        ch = native.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=1)
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        r1 = ch.hmc_run(1e-5, 4, 3)
        if timed:
            us = ch.fused_burst_us(7)
            assert 0.5 < us < 5e4, us
        r2 = ch.hmc_run(1e-5, 4, 3)
        recs.append(([x["log_accept_ratio"] for x in r1 + r2], ch.get_state()))
        ch.close()
    assert recs[0][0] == recs[1][0]
    np.testing.assert_array_equal(recs[0][1], recs[1][1])
