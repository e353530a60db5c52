import sys, os
root = '/root/repo'
for p in (root, root + '/oracle', root + '/tests'): sys.path.insert(0, p)
os.environ["TBNN_JIT_SKIP"] = "fast3,fast,mid,tall"
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
import test_gpu_widefanin as T
from test_gpu_freerun import layers_of
for name in sys.argv[1:]:
    spec, X, Y, theta, eta = T.problem(name)
    lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
    lp32, g32 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float32)[:2]
    for jit in (True, False):
        ch = nat.Chain(layers_of(spec), likelihood=spec.likelihood, fixed_sd=spec.fixed_sd, jit=jit)
        ch.set_data(X, Y)
        lp, g, st = ch.logp_grad(theta, eta)
        print(name, ch.kernel_name, "logp rel", abs(lp - lp64) / abs(lp64))
        for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
            for nm, a, b in (("W", ow, ob), ("b", ob, ob + l.out_dim)):
                sc = max(np.abs(g64[a:b]).max(), 1e-3)
                e64, e32, d = np.abs(g[a:b] - g64[a:b]).max() / sc, np.abs(g[a:b] - g32[a:b]).max() / sc, np.abs(g32[a:b] - g64[a:b]).max() / sc
                print(f"   layer {li} {nm}: vs fp64 {e64:.2e}, vs fp32 oracle {e32:.2e} (fp32 oracle vs fp64 {d:.2e})")
                if nm == "W" and e64 > 1e-4:
                    D = np.abs(g[a:b] - g64[a:b]).reshape(l.out_dim, l.in_dim) / sc
                    r, c = np.unravel_index(np.argmax(D), D.shape)
                    print("      worst at out", r, "in", c, "; rows with err > 1e-4:", np.where(D.max(1) > 1e-4)[0][:20], "cols:", np.where(D.max(0) > 1e-4)[0][:20])
        ch.close()
