"""dev: several independent chains on ONE GPU from one process -- one host thread and one stream per chain (ctypes releases the
GIL inside tbnn_hmc_run): aggregate leapfrog steps/s for 1, 2, 4, 8, 16 chains.  python tools/experiments/multichain_threads.py c1|c2"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tensorbnn_amd import _native as nat
from tensorbnn_amd.workloads import WORKLOADS, burned_state, synth_problem

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
wl = WORKLOADS[name]
layers, lik, X, Y, theta0, eta0 = synth_problem(wl["dims"], wl["n"], prior=wl["prior"], likelihood=wl["lik"])
b = burned_state(name, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
theta0, eta0, eps = b["theta"].astype(np.float32), b["eta"].astype(np.float32), float(b["eps"])
dX, dY = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
L, EP = wl["L"], (200 if name == "c1" else 40)
for C in (1, 2, 4, 8, 16):
    chains = []
    for c in range(C):
        ch = nat.Chain(layers, likelihood=lik, device=0, seed=50, chain_id=c)
        ch.set_data_device(dX.data_ptr(), dY.data_ptr(), wl["n"]); ch.set_state(theta0); ch.set_hypers(eta0)
        ch.hmc_run(eps, L, 2)
        chains.append(ch)
    acc = [None] * C
    def work(i):
        outs = chains[i].hmc_run(eps, L, EP)
        acc[i] = float(np.mean([o["accept_prob"] for o in outs]))
    th = [threading.Thread(target=work, args=(i,)) for i in range(C)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{name}: {C:2d} chains (threads): {C * EP * L / dt:10.0f} leapfrog steps/s aggregate, {EP * L / dt:9.0f} per chain, accept {np.mean(acc):.3f}", flush=True)
    for ch in chains: ch.close()
