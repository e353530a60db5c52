"""dev: random narrow architectures (fan-in <= 16, hidden widths <= 64, 1-2 outputs) through the run-time instantiation of the narrow
family (fast3, else fast), random row counts over a small forced grid (rounds + 0 / 1 / 2 cooperative tiles, ragged last tile), value /
gradient / forward against the fp64 oracle:  python tools/experiments/narrow_fuzz.py [n_shapes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
os.environ["TBNN_JIT_SKIP"] = "mid,tall,wide"
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for k in range(N):
    d_in = int(rng.integers(1, 17))
    nh = int(rng.integers(1, 5))
    # widths around the tile boundaries (16 m + 0 .. 4) as often as anywhere else
    hidden = [int(rng.choice([rng.integers(2, 65), 16 * rng.integers(1, 4) + rng.integers(0, 5)])) for _ in range(nh)]
    d_out = int(rng.integers(1, 3))
    dims = [d_in] + hidden + [d_out]
    fam = jit.families(dims)
    if not fam:
        print(dims, "no narrow family: skipped"); continue
    act = int(rng.choice([o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU]))
    lik = int(rng.choice([o.LIK_GAUSSIAN, o.LIK_BERNOULLI]))
    prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    grid = int(rng.integers(2, 9)); W = 4 * grid
    n = 16 * (int(rng.integers(0, 4)) * W + int(rng.integers(0, 2 * grid + 3))) + int(rng.integers(1, 17))
    os.environ["TBNN_FAST_GRID"] = str(grid)
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    t = time.time()
    try:
        ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=True)
    except Exception as e:
        print(dims, "create failed:", str(e)[:100]); bad += 1; continue
    name = ch.kernel_name
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
    e_g = max(np.abs(g[a:b] - g64[a:b]).max() / max(np.abs(g64[a:b]).max(), 1e-3) for l, (ow, ob) in zip(spec.layers, spec.offsets()) for a, b in ((ow, ob), (ob, ob + l.out_dim)))
    e_f = float(np.abs(f - f64).max())
    ok = e_lp <= 4e-6 and e_g <= 1e-4 and e_f <= 1e-4
    note = ""
    if not ok:
        # a saturated Bernoulli problem (outputs within 1e-7 of 0 / 1) is ill-conditioned in fp32: the fp32 ORACLE is then the reference
        lp32, g32 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float32)[:2]
        e_lp32 = abs(lp - lp32) / max(abs(lp64), 1.0)
        e_g32 = float(np.abs(g - g32).max() / max(np.abs(g64).max(), 1e-3))
        d32 = float(np.abs(g32 - g64).max() / max(np.abs(g64).max(), 1e-3))
        if e_g32 <= 2e-6 and e_lp32 <= 4e-6 and e_f <= 1e-4:
            ok = True; note = f" [ill-conditioned problem: fp32 oracle is {d32:.1e} from fp64, kernel {e_g32:.1e} from the fp32 oracle]"
    if not ok and act == o.ACT_RELU:
        kk = relu_kink(spec, theta, X)
        if kk < 3e-6 and e_lp <= 4e-6 and e_f <= 1e-4 and e_g <= 20.0 / max(n, 1):
            ok = True; note = f" [a relu pre-activation within fp32 rounding of 0 ({kk:.1e} of the layer's scale): one derivative flips, gradient {e_g:.1e} ~ 1 / rows]"
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {dims} n={n} grid={grid} act={act} lik={lik} prior={prior}: {name}; logp {e_lp:.1e} grad {e_g:.1e} forward {e_f:.1e} ({time.time() - t:.0f} s){note}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
