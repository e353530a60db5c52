"""Generates tests/golden/*.npz from the CPU oracle (oracle/tbnn_oracle.py).

The reference ships no golden vectors and cannot be imported here (TensorFlow /
TFP absent), so these fixtures come from the repo's own restatement
("parity unpinned").  They freeze the oracle's outputs so that (a) a
later change of the oracle is visible and (b) the GPU parity tests have fixed
inputs/outputs that travel to the GPU box.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import tbnn_oracle as o  # noqa: E402

CASES = {
    "c1": dict(dims=[1, 10, 10, 1], n=256, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, eps=2e-4, eps_h=1e-5),
    "trainreg": dict(dims=[1, 10, 10, 10, 1], n=11, act=o.ACT_TANH, prior=o.PRIOR_GAUSSIAN, lik=o.LIK_FIXED_GAUSSIAN, eps=2e-3, eps_h=1e-4),
    "c2": dict(dims=[5, 50, 50, 50, 1], n=256, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_GAUSSIAN, eps=5e-5, eps_h=1e-5),
    "c5": dict(dims=[20, 100, 100, 2], n=256, act=o.ACT_RELU, prior=o.PRIOR_CAUCHY, lik=o.LIK_BERNOULLI, eps=2e-4, eps_h=1e-4),
}


def main():
    for name, c in CASES.items():
        spec, X, Y, theta, eta = o.synth_problem(c["dims"], c["n"], c["act"], c["prior"], c["lik"])
        rng = np.random.default_rng(2024)
        p0 = rng.standard_normal(spec.n_params).astype(np.float32)
        hp0 = rng.standard_normal(spec.n_hypers).astype(np.float32)
        out = dict(dims=np.array(c["dims"]), act=c["act"], prior=c["prior"], lik=c["lik"], X=X, Y=Y, theta=theta, eta=eta,
                   p0=p0, hp0=hp0, eps=c["eps"], eps_h=c["eps_h"])
        f64, acts = o.forward(spec, theta, X, np.float64, keep=True)
        out["forward64"] = f64
        for tag, dt in (("32", np.float32), ("64", np.float64)):
            lp, g = o.target_log_prob_and_grad(spec, theta, eta, X, Y, dt)
            out["logp" + tag], out["grad" + tag] = np.float64(lp), g.astype(np.float64)
            hlp, hg = o.hyper_log_prob_and_grad(spec, eta, theta, X, Y, dt)
            out["hyper_logp" + tag], out["hyper_grad" + tag] = np.float64(hlp), hg.astype(np.float64)
        # 5-step leapfrog trajectory + accept decision, injected p0 / log u
        for lu_tag, lu in (("half", np.log(0.5)), ("acc", -1e30), ("rej", 1e30)):
            r = o.weight_step(spec, theta, eta, X, Y, c["eps"], 5, p0, lu, np.float64)
            out[f"step_{lu_tag}_trace"] = np.array(r.trace_logp)
            out[f"step_{lu_tag}_lar"] = r.log_accept_ratio
            out[f"step_{lu_tag}_accepted"] = r.accepted
            out[f"step_{lu_tag}_theta"] = r.theta
            out[f"step_{lu_tag}_sjd"] = r.sjd
        r32 = o.weight_step(spec, theta, eta, X, Y, c["eps"], 5, p0, np.log(0.5), np.float32)
        out["step_half_lar32"] = r32.log_accept_ratio
        # hyper step + 20 epochs of dual averaging (log-accept ratios from repeated hyper steps)
        hs = o.hyper_step(spec, eta, theta, X, Y, c["eps_h"], 20, hp0, -1e30, np.float64)
        out["hyper_step_lar"], out["hyper_step_eta"] = hs.log_accept_ratio, hs.theta
        st = o.DualAveragingState(hyper_step_size=0.01, burnin=100)
        lars = rng.normal(-0.5, 1.0, 20)
        da = []
        for ep, lar in enumerate(lars):
            acc = o.dual_averaging_update(st, ep, lar)
            da.append((acc, st.h, st.log_eps_bar, st.eps_h))
        out["da_lars"], out["da_trace"] = lars, np.array(da)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, "P", spec.n_params, "logp64", out["logp64"], "lar", out["step_half_lar"], "lar32", out["step_half_lar32"])

    # density known answers (reference formula; scipy deltas are asserted in tests/test_oracle.py)
    x = np.linspace(-3, 3, 13)
    np.savez_compressed(os.path.join(HERE, "density_kat.npz"), x=x,
                        cauchy=o.cauchy_log_prob(0.5, 0.1, x, np.float64),
                        mvn_vec=o.multivariate_log_prob(np.full(13, 0.7), 0.2, x, np.float64),
                        mvn_scalar_sigma=o.multivariate_log_prob(0.7, 0.2, x, np.float64))

    # adapter trace: 60 update() calls with injected uniforms / grid picks
    rng = np.random.default_rng(5)
    P, n_upd = 141, 60
    uniforms = rng.random(n_upd).astype(np.float32)
    ce, cl = rng.integers(0, 40, n_upd), rng.integers(0, 91, n_upd)
    ad = o.ParamAdapter(1e-3, 100, 1e-4, 1e-2, 40, 10, 100, 1, 2, 5, a=4, delta=0.1, randomSteps=3)
    state = rng.standard_normal(P).astype(np.float32)
    states, outs = [], []
    for t in range(n_upd):
        step = float(ad.currentE) * np.sqrt(float(ad.currentL)) * 30
        state = (state + step * rng.standard_normal(P).astype(np.float32) * (rng.random() < 0.8)).astype(np.float32)
        ad._uniforms, ad._choices = [uniforms[t]], [int(ce[t]), int(cl[t])]
        e, L = ad.update(state.copy())
        states.append(state.copy())
        outs.append((float(e), int(L)))
    np.savez_compressed(os.path.join(HERE, "adapter_trace.npz"), states=np.array(states), uniforms=uniforms, ce=ce, cl=cl,
                        outs=np.array(outs), sjd=np.array(ad.sjd_log),
                        ctor=np.array([1e-3, 100, 1e-4, 1e-2, 40, 10, 100, 1, 2, 5, 4, 0.1, 3]))
    print("adapter trace: grid search used:", ad.i // ad.m >= ad.randomSteps, "history", len(ad.previousGamma))


if __name__ == "__main__":
    main()
