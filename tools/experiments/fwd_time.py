"""dev: forward-only throughput (tbnn_forward_many: m networks x n rows) of a shape:  python tools/experiments/fwd_time.py 784,20,20,1 60000 16 bern"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import tbnn_oracle as o
from tensorbnn_amd import _native as nat
dims = [int(x) for x in sys.argv[1].split(",")]; n = int(sys.argv[2]); m = int(sys.argv[3])
lik = o.LIK_BERNOULLI if len(sys.argv) > 4 and sys.argv[4] == "bern" else o.LIK_GAUSSIAN
spec, X, Y, theta, eta = o.synth_problem(dims, n, o.ACT_RELU, o.PRIOR_CAUCHY, lik)
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood)
ch.set_data(X, Y); ch.set_validation(X, Y)
thetas = np.tile(theta, (m, 1)) + 0.01 * np.random.default_rng(0).standard_normal((m, theta.size)).astype(np.float32)
out = ch.forward_many(thetas, which=1)
ref = o.forward(spec, thetas[m - 1], X[:512], np.float64)
print(ch.kernel_name, "max |f - f64|", float(np.abs(out[m - 1][:, :512] - ref).max()))
t = time.perf_counter()
for _ in range(5): ch.forward_many(thetas, which=1)
dt = (time.perf_counter() - t) / 5
print(f"forward_many {m} networks x {n} rows: {dt * 1e3:.2f} ms  ({m * n / dt / 1e6:.1f} M rows/s incl. the copy of the outputs)")
