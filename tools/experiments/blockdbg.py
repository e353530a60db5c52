"""dev: one shape, four launches: per (layer, W | b) block the gradient error against the fp64 and the fp32 oracle, the 16 x 16 tiles that are
off, and whether a repeated launch is bit-identical:  python tools/experiments/blockdbg.py <dims> <rows> <act> <lik> <prior> [TBNN_JIT_SKIP list] [theta scale]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import tbnn_oracle as o
dims = [int(v) for v in sys.argv[1].split(",")]; n = int(sys.argv[2]); act = int(sys.argv[3]); lik = int(sys.argv[4]); prior = int(sys.argv[5])
if len(sys.argv) > 6 and sys.argv[6] != "-": os.environ["TBNN_JIT_SKIP"] = sys.argv[6]
from tensorbnn_amd import _native as nat
spec, X, Y, theta, eta = o.synth_problem(dims, n, act, prior, lik)
if dims[0] > 64: X = (X / np.sqrt(dims[0] / 16.0)).astype(np.float32)          # (as family_fuzz.py / transition_fuzz.py)
if len(sys.argv) > 7: theta = (theta * float(sys.argv[7])).astype(np.float32)
lp64, g64 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float64)[:2]
lp32, g32 = o.target_log_prob_and_grad(spec, theta, eta, X, Y, np.float32)[:2]
ch = nat.Chain([(l.in_dim, l.out_dim, l.act, l.prior) for l in spec.layers], likelihood=spec.likelihood, jit=True)
print(ch.kernel_name)
ch.set_data(X, Y)
prev = None
for rep in range(4):
    lp, g, st = ch.logp_grad(theta, eta)
    print("launch", rep, "logp", lp, "(fp64", lp64, "fp32", float(lp32), ") finite", bool(np.isfinite(g).all()), "same as the launch before:", None if prev is None else bool(np.array_equal(prev, g, equal_nan=True)))
    prev = g
    if rep not in (0, 3): continue
    for li, (l, (ow, ob)) in enumerate(zip(spec.layers, spec.offsets())):
        for nm, a, b in (("W", ow, ob), ("b", ob, ob + l.out_dim)):
            sc = max(np.abs(g64[a:b]).max(), 1e-3)
            e64, e32, d = np.abs(g[a:b] - g64[a:b]).max() / sc, np.abs(g[a:b] - g32[a:b]).max() / sc, np.abs(g32[a:b] - g64[a:b]).max() / sc
            if e64 <= 1e-4: continue
            print(f"   layer {li} {nm}: vs fp64 {e64:.2e}, vs fp32 oracle {e32:.2e} (fp32 oracle vs fp64 {d:.2e})")
            if nm == "b" and e32 > 1e-5:
                bad = np.nonzero(~(np.abs(g[a:b] - g32[a:b]) <= 1e-5 * sc))[0]
                print("     off: units", bad.tolist()[:20], "kernel", g[a:b][bad][:6].tolist(), "oracle", g32[a:b][bad][:6].tolist())
            if nm == "W" and e32 > 1e-5:
                Wg = g[a:b].reshape(l.out_dim, l.in_dim); W0 = g32[a:b].reshape(l.out_dim, l.in_dim)
                badm = ~(np.abs(Wg - W0) <= 1e-5 * sc); rows, cols = np.nonzero(badm)
                print("     off:", int(badm.sum()), "entries; M tiles", sorted(set((rows // 16).tolist())), "N tiles", sorted(set((cols // 16).tolist())))
ch.close()
