"""dev: score identities E[g] = 0, E[theta*g] = -1 on samples of the fused kernels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
def bse(x, nb=40):
    T = (x.shape[0] // nb) * nb
    bm = x[:T].reshape(nb, T // nb, -1).mean(axis=1)
    return bm.std(axis=0, ddof=1) / np.sqrt(nb)
for act, name in ((o.ACT_RELU, "relu"), (o.ACT_TANH, "tanh")):
    dims = [1, 10, 10, 1] if act == o.ACT_RELU else [1, 10, 10, 10, 1]
    spec, X, Y, theta, eta = o.synth_problem(dims, 256, act, o.PRIOR_GAUSSIAN, o.LIK_FIXED_GAUSSIAN)
    layers = [(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers]
    ch = nat.Chain(layers, likelihood=spec.likelihood, fixed_sd=0.3, seed=5, chain_id=1)
    ch.set_data(X, Y)
    eta = np.tile(np.array([0.0, 1.0, 0.0, 1.0], np.float32), len(layers))
    ch.set_hypers(eta); ch.set_state((0.3 * theta).astype(np.float32))
    L, eps = 200, 1e-3
    ch.hmc_run(eps, L, 300)
    for _ in range(30):
        acc = np.mean([o_["accept_prob"] for o_ in ch.hmc_run(eps, L, 100)])
        if acc < 0.65: eps *= 0.8
        elif acc > 0.9: eps *= 1.2
        else: break
    ch.hmc_run(eps, L, 1000)
    T, P = 1500, ch.P
    G = np.empty((T, P)); TH = np.empty((T, P)); acc = []; LP = np.empty(T)
    for t in range(T):
        acc.append(ch.hmc_step(eps, L)["accept_prob"])
        th = ch.get_state(); lp, g, _ = ch.logp_grad(th, eta)
        TH[t] = th; G[t] = g; LP[t] = lp
    z1 = G.mean(axis=0) / bse(G); v = TH * G; z2 = (v.mean(axis=0) + 1.0) / bse(v)
    print(name, ch.kernel_name, "eps", eps, "acc", np.mean(acc), "logp first/last quarter", LP[:T//4].mean(), LP[-T//4:].mean())
    print("  z1: max", np.abs(z1).max(), "mean z^2", np.mean(z1**2), "argmax", np.argmax(np.abs(z1)), " mean|g|", np.abs(G.mean(0)).max(), "typ |g| sd", G.std(0).mean())
    print("  z2: max", np.abs(z2).max(), "mean z^2", np.mean(z2**2), "argmax", np.argmax(np.abs(z2)), " E[theta g] range", v.mean(0).min(), v.mean(0).max())
    ch.close()
