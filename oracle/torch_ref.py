"""PyTorch-CPU value-and-gradient of the weight target -- TEST INFRASTRUCTURE ONLY (the second cpu_baseline entry of
bench.py, BASELINE.md section 4 "secondary cross-check", and an autograd check of the oracle in tests/test_oracle.py).

The closure calculateProbs of InnerStepMain (network.py:370-392) written with torch ops in float32, differentiated by
torch.autograd the way the reference lets TensorFlow differentiate it: dense forward layer.py:266-279, activations
activationFunctions.py:27-63, Gaussian / Bernoulli likelihood likelihood.py:69-96, :210-237, Cauchy / Gaussian weight
priors layer.py:166-197, :346-377 with the quirks of BNN_functions.py:7-57 (Q1 sign, Q2 normaliser, Q3 squares).
No graph is re-used between calls (a fresh autograd tape per evaluation, like one TF eager value_and_gradient)."""
import math

import numpy as np
import torch

import tbnn_oracle as o


class TorchTarget:
    def __init__(self, spec, X, Y, dtype=torch.float32, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.spec, self.dt = spec, dtype
        self.Xt = torch.from_numpy(np.ascontiguousarray(X)).to(dtype).T.contiguous()          # [d_in, n] (network.py:41-44)
        self.Y = torch.from_numpy(np.ascontiguousarray(Y)).to(dtype).reshape(X.shape[0], -1).T.contiguous()
        self.threads = torch.get_num_threads()

    def _prior(self, l, h4, W, b):
        tot = 0.0
        for x, loc, g in ((W, h4[0], h4[1]), (b, h4[2], h4[3])):
            sc = g * g                                                                         # Q3
            if l.prior == o.PRIOR_CAUCHY:
                z = (x - loc) / sc
                tot = tot + torch.sum(torch.log1p(z * z) - math.log(math.pi) - torch.log(sc))  # Q1: + log(1+z^2)
            else:
                s = torch.clamp(sc, 1e-8, 1e8)
                tot = tot - 0.5 * (2.0 * torch.log(s) + torch.sum(((x - loc) / s) ** 2) + math.log(2 * math.pi))   # Q2: k = 1
        return tot

    def value_and_grad(self, theta, eta):
        spec, dt = self.spec, self.dt
        th = torch.tensor(np.asarray(theta), dtype=dt, requires_grad=True)
        et = torch.tensor(np.asarray(eta), dtype=dt)
        a, off, tot = self.Xt, 0, torch.zeros((), dtype=dt)
        acts = {o.ACT_NONE: lambda v: v, o.ACT_RELU: torch.relu, o.ACT_TANH: torch.tanh, o.ACT_SIGMOID: torch.sigmoid,
                o.ACT_EXP: torch.exp, o.ACT_ELU: torch.nn.functional.elu}
        for i, l in enumerate(spec.layers):
            W = th[off:off + l.in_dim * l.out_dim].reshape(l.out_dim, l.in_dim); off += l.in_dim * l.out_dim
            b = th[off:off + l.out_dim].reshape(l.out_dim, 1); off += l.out_dim
            tot = tot + self._prior(l, et[4 * i:4 * i + 4], W, b)
            a = acts[l.act](W @ a + b)
        if spec.likelihood == o.LIK_BERNOULLI:
            p = torch.clamp(a, 1e-8, 1 - 1e-7)                                                # likelihood.py:226-231
            tot = tot + torch.sum(torch.xlogy(self.Y, p) + torch.xlogy(1 - self.Y, 1 - p))
        else:
            s = float(np.clip(float(eta[-1]) ** 2 if spec.likelihood == o.LIK_GAUSSIAN else spec.fixed_sd, 1e-8, 1e8))
            tot = tot - 0.5 * (2 * a.numel() * math.log(s) + torch.sum((self.Y - a) ** 2) / (s * s) + a.numel() * math.log(2 * math.pi))
        g, = torch.autograd.grad(tot, th)
        return float(tot.detach()), g.numpy()
