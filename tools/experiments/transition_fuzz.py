"""dev: random architectures (whatever family serves them), one injected HMC transition and one hyper transition against the fp64 oracle, and a
chain group of three against its solo chains bit for bit (free-running on the device's draws):
  python tools/experiments/transition_fuzz.py [n_shapes] [seed] [family 0..4: narrow, mid-width, tall, wide, anything; default: all] [max rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
REC = ("log_accept_ratio", "accepted", "logp_old", "logp_new", "kinetic_old", "kinetic_new", "sjd", "accept_prob")
bad = 0
for k in range(N):
    kind = int(sys.argv[3]) if len(sys.argv) > 3 else int(rng.integers(0, 5))
    if kind == 0: dims = [int(rng.integers(1, 17))] + [int(rng.integers(2, 65)) for _ in range(int(rng.integers(1, 4)))] + [int(rng.integers(1, 3))]          # narrow
    elif kind == 1: dims = [int(rng.integers(1, 100))] + [int(rng.integers(17, 112)) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]     # mid-width
    elif kind == 2: dims = [int(rng.integers(33, 900))] + [int(rng.integers(3, 65)) for _ in range(int(rng.integers(1, 3)))] + [int(rng.integers(1, 3))]      # tall
    elif kind == 3: dims = [int(rng.integers(1, 33))] + [int(rng.integers(65, 257)) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]      # wide
    else: dims = [int(rng.integers(1, 400))] + [int(rng.integers(2, 300)) for _ in range(int(rng.integers(1, 4)))] + [int(rng.choice([1, 3, 7]))]            # anything
    act = int(rng.choice([o.ACT_RELU, o.ACT_TANH, o.ACT_SIGMOID, o.ACT_ELU]))
    lik = int(rng.choice([o.LIK_GAUSSIAN, o.LIK_GAUSSIAN, o.LIK_BERNOULLI])) if dims[-1] <= 2 else o.LIK_GAUSSIAN
    prior = int(rng.choice([o.PRIOR_CAUCHY, o.PRIOR_GAUSSIAN]))
    n = int(rng.choice([rng.integers(1, 64), rng.integers(64, int(sys.argv[4]) if len(sys.argv) > 4 else 2500)]))
    spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
    if dims[0] > 64: X = (X / np.sqrt(dims[0] / 16.0)).astype(np.float32)
    if lik == o.LIK_BERNOULLI: theta = (theta * 0.3).astype(np.float32)          # keep the outputs off saturation: a well-conditioned fp32 problem
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    L = int(rng.integers(1, 6)); eps = float(10.0 ** rng.uniform(-6.0, -4.5))
    p0 = rng.standard_normal(spec.n_params).astype(np.float32)
    t = time.time()
    msgs = []
    try:
        ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=2, jit=True)
        name = ch.kernel_name
        ch.set_data(X, Y); ch.set_state(theta); ch.set_hypers(eta)
        lp64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[0]
        for log_u in (-1e30, 1e30):
            ch.set_state(theta); ch.set_hypers(eta)
            out = ch.hmc_step(eps, L, p0=p0, log_u=log_u)
            ref = o.weight_step(spec, theta, eta, X, Y, eps, L, p0, log_u, np.float64)
            tol = 2e-2 + 1e-4 * abs(ref.log_accept_ratio) + 4e-7 * abs(lp64)
            if not abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= tol: msgs.append(f"lar {out['log_accept_ratio']:.5g} vs {ref.log_accept_ratio:.5g} (tol {tol:.2g})")
            if bool(out["accepted"]) != ref.accepted: msgs.append("decision")
            d = np.abs(ch.get_state() - ref.theta).max()
            if not d <= 1e-5 * max(1.0, np.abs(ref.theta).max()): msgs.append(f"state {d:.2e}")
        if spec.n_hypers:
            ph = rng.standard_normal(spec.n_hypers).astype(np.float32)
            ch.set_state(theta); ch.set_hypers(eta)
            ch.logp_grad(theta, eta)                       # the cached statistic the hyper target uses
            out = ch.hyper_step(1e-4, 9, p0=ph, log_u=-1e30)
            ref = o.hyper_step(spec, eta, theta, X, Y, 1e-4, 9, ph, -1e30, np.float64)
            if not abs(out["log_accept_ratio"] - ref.log_accept_ratio) <= 2e-2 + 1e-3 * abs(ref.log_accept_ratio): msgs.append(f"hyper lar {out['log_accept_ratio']:.5g} vs {ref.log_accept_ratio:.5g}")
            if not np.allclose(ch.get_hypers(), ref.theta, rtol=1e-4, atol=1e-5): msgs.append("hyper state")
        ch.close()
        # a group of three on the device's own draws against its solo chains
        C = 3
        thetas = (theta[None, :] * (1.0 + 0.03 * rng.standard_normal((C, theta.size)))).astype(np.float32)
        etas = np.tile(eta, (C, 1)).astype(np.float32)
        grp = nat.ChainGroup(layers, C, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=4, jit=True)
        grp.set_data(X, Y); grp.set_state(thetas); grp.set_hypers(etas)
        g1 = grp.hmc_run(eps, L, 3)
        gs = grp.get_state(); grp.close()
        for c in range(C):
            s = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, seed=50, chain_id=4 + c, jit=True)
            s.set_data(X, Y); s.set_state(thetas[c]); s.set_hypers(etas[c])
            s1 = s.hmc_run(eps, L, 3)
            for got, want in zip(g1[c], s1):
                if not all(np.float64(got[q]) == np.float64(want[q]) or (np.isnan(got[q]) and np.isnan(want[q])) for q in REC): msgs.append(f"group chain {c} record"); break
            if not np.array_equal(gs[c], s.get_state()): msgs.append(f"group chain {c} state")
            s.close()
    except Exception as e:
        msgs.append("exception: " + str(e)[:160]); name = "?"
    bad += bool(msgs)
    print(f"{'BAD' if msgs else 'ok '} {dims} n={n} act={act} lik={lik} prior={prior} L={L} eps={eps:.1e}: {name} ({time.time() - t:.0f} s) {'; '.join(msgs)}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
