"""dev: aggregate leapfrog steps/s of a chain group (tbnn_create_multi) against the solo chain:  python tools/experiments/multichain_group.py c1|c2 [chains...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import WORKLOADS, burned_state, synth_problem

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64]
wl = WORKLOADS[name]
layers, lik, X, Y, theta0, eta0 = synth_problem(wl["dims"], wl["n"], prior=wl["prior"], likelihood=wl["lik"], x_scale=wl.get("x_scale"))
b = burned_state(name, os.path.join(ROOT, "tests", "golden"))
theta0, eta0, eps = b["theta"].astype(np.float32), b["eta"].astype(np.float32), float(b["eps"])
dX, dY = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
L, EP = wl["L"], (200 if name == "c1" else 40)
ch = nat.Chain(layers, likelihood=lik)
ch.set_data_device(dX.data_ptr(), dY.data_ptr(), wl["n"]); ch.set_state(theta0); ch.set_hypers(eta0)
ch.hmc_run(eps, L, 5); torch.cuda.synchronize()
t0 = time.perf_counter(); outs = ch.hmc_run(eps, L, EP); dt = time.perf_counter() - t0
print(f"{name}: solo Chain {EP * L / dt:10.0f} leapfrog steps/s, accept {np.mean([o['accept_prob'] for o in outs]):.3f}  ({ch.kernel_name})", flush=True)
ch.close()
for C in counts:
    g = nat.ChainGroup(layers, C, likelihood=lik)
    g.set_data_device(dX.data_ptr(), dY.data_ptr(), wl["n"]); g.set_state(theta0); g.set_hypers(eta0)
    g.hmc_run(eps, L, 5); torch.cuda.synchronize()
    t0 = time.perf_counter(); outs = g.hmc_run(eps, L, EP); dt = time.perf_counter() - t0
    acc = np.mean([o["accept_prob"] for c in outs for o in c])
    print(f"{name}: group of {C:3d}: {C * EP * L / dt:10.0f} leapfrog steps/s aggregate, {EP * L / dt:9.0f} per chain, accept {acc:.3f}", flush=True)
    g.close()
