"""dev: random architectures through ONE kernel family (mid | wide | layered), random row counts, activations, likelihoods, priors:
value / gradient / forward against the fp64 oracle (saturated Bernoulli problems against the fp32 oracle, as narrow_fuzz.py):
  python tools/experiments/family_fuzz.py mid|wide|layered [n_shapes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
FAM = sys.argv[1]
os.environ["TBNN_JIT_SKIP"] = {"mid": "fast3,fast,tall,wide", "wide": "fast3,fast,tall,mid", "layered": "fast3,fast,tall,mid,wide"}[FAM]
if FAM == "layered": os.environ["TBNN_TALL"] = "0"
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat, jit

def relu_kink(spec, theta, X):
    """smallest |pre-activation| of a relu hidden layer relative to the layer's mean |z| (fp64): below ~1e-6 the fp32 sign of that z depends on
    the summation order, and one (row, unit) whose relu derivative flips moves the gradient by O(1 / rows) -- the problem, not the kernel"""
    a = X.astype(np.float64); worst = np.inf
    for l, (ow, ob) in list(zip(spec.layers, spec.offsets()))[:-1]:
        z = a @ theta[ow:ob].reshape(l.out_dim, l.in_dim).astype(np.float64).T + theta[ob:ob + l.out_dim].astype(np.float64)
        if l.act == o.ACT_RELU: worst = min(worst, np.abs(z).min() / max(np.abs(z).mean(), 1e-30))
        a = o.act_forward(l.act, z) if hasattr(o, "act_forward") else (np.maximum(z, 0) if l.act == o.ACT_RELU else np.tanh(z) if l.act == o.ACT_TANH else 1 / (1 + np.exp(-z)) if l.act == o.ACT_SIGMOID else np.where(z > 0, z, np.exp(np.minimum(z, 0)) - 1))
    return worst
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 7)
bad = done = 0
tries = 0
while done < N and tries < 40 * N:
    tries += 1
    edge = lambda lo, hi: int(rng.choice([rng.integers(lo, hi + 1), min(hi, 16 * rng.integers(max(1, lo // 16), hi // 16 + 1) + rng.integers(0, 6))]))
    if FAM == "mid":
        dims = [edge(1, 128)] + [edge(17, 112) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]
    elif FAM == "wide":
        dims = [edge(1, 32)] + [edge(65, 256) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]
    else:
        dims = [edge(1, 900)] + [edge(2, 320) for _ in range(int(rng.integers(1, 4)))] + [int(rng.choice([1, 2, 3, 5, 10, 17]))]
    fam = jit.families(dims)
    if FAM != "layered" and FAM not in fam:
        continue
    act = int(rng.choice([o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU]))
    lik = int(rng.choice([o.LIK_GAUSSIAN, o.LIK_BERNOULLI]))
    prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 30000)]))
    if sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) * n > 4e9: n = max(16, n // 8)
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    if dims[0] > 64: X = (X / np.sqrt(dims[0] / 16.0)).astype(np.float32)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    t = time.time()
    try:
        ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True)
    except Exception as e:
        print(dims, "create failed:", str(e)[:120]); bad += 1; done += 1; continue
    name = ch.kernel_name
    if FAM != "layered" and FAM not in name:
        print(dims, "took", name, "(the family's instantiation did not build: skipped)"); ch.close(); continue
    done += 1
    ch.set_data(X, Y)
    lp, g, st = ch.logp_grad(theta, eta)
    # an ensemble of three networks through the forward-only kernels (blockIdx.y = network), the largest deviation of the three
    ths = np.stack([theta, (theta * 0.9).astype(np.float32), (theta * 1.07).astype(np.float32)])
    fm = ch.forward_many(ths, X=X[: min(n, 500)])
    f = fm[0]
    f64 = o.forward(spec, theta, X[: min(n, 500)], np.float64)
    for kk in (1, 2):
        dk = fm[kk] - o.forward(spec, ths[kk], X[: min(n, 500)], np.float64)
        if np.abs(dk).max() > np.abs(f - f64).max(): f, f64 = fm[kk], fm[kk] - dk
    ch.close()
    e_lp = abs(lp - lp64) / max(abs(lp64), 1.0)
    blocks = [(a, b) for l, (ow, ob) in zip(spec.layers, spec.offsets()) for a, b in ((ow, ob), (ob, ob + l.out_dim))]
    e_g = max(np.abs(g[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for a, b in blocks)
    e_f = float(np.abs(f - f64).max())
    ok = e_lp <= 4e-6 and e_g <= 1e-4 and e_f <= 1e-4
    note = ""
    if not ok:
        lp32, g32 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float32)[:2]
        e_lp32 = abs(lp - lp32) / max(abs(lp64), 1.0)
        e_g32 = max(np.abs(g[a:b] - g32[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for a, b in blocks)
        d32 = max(np.abs(g32[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for a, b in blocks)
        if e_g32 <= max(3e-6, 0.02 * d32) and e_lp32 <= 4e-6 and e_f <= 1e-4:
            ok = True; note = f" [ill-conditioned problem: fp32 oracle {d32:.1e} from fp64, kernel {e_g32:.1e} from the fp32 oracle]"
    if not ok and act == o.ACT_RELU:
        kk = relu_kink(spec, theta, X)
        if kk < 3e-6 and e_lp <= 4e-6 and e_f <= 1e-4 and e_g <= 20.0 / max(n, 1):
            ok = True; note = f" [a relu pre-activation within fp32 rounding of 0 ({kk:.1e} of the layer's scale): one derivative flips, gradient {e_g:.1e} ~ 1 / rows]"
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {dims} n={n} act={act} lik={lik} prior={prior}: {name}; logp {e_lp:.1e} grad {e_g:.1e} forward {e_f:.1e} ({time.time() - t:.0f} s){note}", flush=True)
print("shapes:", done, "failures:", bad)
sys.exit(1 if bad else 0)
