import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""dev: step-size scan for the C4 / C5 bench workloads (accept-prob window [0.6, 0.9] after warm-up)"""
import sys, json, numpy as np
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import synth_problem
case = sys.argv[1]
dims, n, lik, L, eps_w, cands, nw, nt = {
    "c4": ([10, 200, 200, 200, 1], 1_000_000, nat.LIK_GAUSSIAN, 100, 1e-6, (1.6e-5, 3.2e-5, 6.4e-5), 20, 20),
    "c5": ([20, 100, 100, 2], 500_000, nat.LIK_BERNOULLI, 50, 5e-5, (2.5e-5, 5e-5, 1e-4, 2e-4, 4e-4), 20, 60),
}[case]
layers, lik, X, Y, th, eta = synth_problem(dims, n, likelihood=lik)
res = {}
for eps in cands:
    ch = nat.Chain(layers, likelihood=lik); ch.set_data(X, Y); ch.set_state(th); ch.set_hypers(eta)
    w = ch.hmc_run(eps_w, L, nw)
    o = ch.hmc_run(eps, L, nt)
    ap = np.array([x['accept_prob'] for x in o]); ac = np.array([x['accepted'] for x in o])
    h = nt // 2
    res[str(eps)] = [round(float(ap.mean()), 3), [round(float(ap[:h].mean()), 2), round(float(ap[h:].mean()), 2)]]
    print(case, 'eps', eps, 'warm acc', round(float(np.mean([x['accept_prob'] for x in w])), 3), 'timed acc_prob', res[str(eps)],
          'accepted', round(float(ac.mean()), 3), 'logp start/end', round(o[0]['logp_old'], 1), round(o[-1]['logp_old'], 1), flush=True)
    ch.close()
json.dump({"config": case, "L": L, "eps_warm": eps_w, "warmup": nw, "timed": nt, "scan": res}, open(f"gpurun_out/epsscan_{case}.json", "w"), indent=1)
