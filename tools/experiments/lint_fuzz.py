"""dev (no GPU): compile random shapes of one kernel family with jit.build and report what the build-time hazard check (hazard_lint through
checked_compile) did -- clean at once, repaired (how many pairs of which rule got their wait states), or refused (no kernel of the family):
  python tools/experiments/lint_fuzz.py narrow|mid|tall|wide [n_shapes] [seed] [processes]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
FAM = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
NP = int(sys.argv[4]) if len(sys.argv) > 4 else 6
os.environ["TBNN_JIT_SKIP"] = {"narrow": "mid,tall,wide", "mid": "fast3,fast,tall,wide", "tall": "fast3,fast,mid,wide", "wide": "fast3,fast,tall,mid"}[FAM]
os.environ["TBNN_JIT_DIR"] = tempfile.mkdtemp(prefix="tbnn_lintfuzz_")

def one(job):
    dims, act, bern = job
    import io, contextlib, warnings
    from tensorbnn_amd import jit, _native as nat, hazard_lint as hl
    layers = [(dims[i], dims[i + 1], act if i < len(dims) - 2 else (nat.ACT_SIGMOID if bern else nat.ACT_NONE), nat.PRIOR_CAUCHY) for i in range(len(dims) - 1)]
    err = io.StringIO()
    with contextlib.redirect_stderr(err), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        so = jit.build(layers, nat.LIK_BERNOULLI if bern else nat.LIK_GAUSSIAN, verbose=True)
    txt = err.getvalue()
    if so is None:
        fails = [f for f in os.listdir(os.environ["TBNN_JIT_DIR"]) if f.endswith(".fail")]
        why = ""
        for f in fails:
            t = open(os.path.join(os.environ["TBNN_JIT_DIR"], f)).read()
            if str(dims) in t or True:
                m = [l for l in t.splitlines() if "error:" in l]
                why = m[-1][-220:] if m else why
        return dims, "REFUSED", why
    st = jit.lint_status(so)
    import re
    m = re.search(r"listing checked \((\d+) repaired(?:: ([^)]*))?\)", st)
    n = int(m.group(1)) if m else -1
    return dims, ("clean" if n == 0 else "repaired" if n > 0 else "UNKNOWN"), (m.group(2) or "") if m else st

if __name__ == "__main__":
    from tensorbnn_amd import jit
    jobs = []
    while len(jobs) < N:
        if FAM == "narrow": dims = [int(rng.integers(1, 17))] + [int(rng.integers(2, 65)) for _ in range(int(rng.integers(1, 5)))] + [int(rng.integers(1, 3))]
        elif FAM == "mid": dims = [int(rng.integers(1, 129))] + [int(rng.integers(17, 113)) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]
        elif FAM == "tall": dims = [int(rng.integers(33, 1300))] + [int(rng.integers(3, 65)) for _ in range(int(rng.integers(1, 4)))] + [int(rng.integers(1, 3))]
        else: dims = [int(rng.integers(1, 33))] + [int(rng.integers(65, 257)) for _ in range(int(rng.integers(2, 4)))] + [int(rng.integers(1, 3))]
        if not jit.families(dims): continue
        jobs.append((dims, int(rng.choice([1, 2, 3, 5])), bool(rng.integers(0, 2))))
    from multiprocessing import Pool
    with Pool(NP) as pool:
        res = pool.map(one, jobs, chunksize=1)
    tally, rules = {}, {}
    for dims, what, why in res:
        tally[what] = tally.get(what, 0) + 1
        if what != "clean": print(dims, what, why, flush=True)
        for kv in (why.split() if what == "repaired" else []):
            k, v = kv.split(":")
            rules[k] = rules.get(k, 0) + int(v)
    print(FAM, tally, "pairs repaired by rule:", rules)
