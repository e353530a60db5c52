"""Summaries of one tools/profile_round.sh run: per-workload PMC means per launch of the fused-pass kernels, the kernel time
of one fused pass from the rocprofv3 kernel stats (rocprof_kernel_us.json: what bench.py's roofline.frac_rocprof uses) and the
HBM traffic per pass with the gfx950 correction (pmc_traffic.json: roofline.traffic).  python tools/rocprof_summary.py <dir> [tag]"""
import collections, csv, glob, json, os, sys

out = sys.argv[1]
tag = os.path.basename(os.path.normpath(out))
# kernels of one fused forward+backward pass, per workload
PASS = {"c2": ["k_fwd_bwd_fast3"], "c4": ["k_chain_wide", "k_dw_wide", "k_reduce_wide"], "c5": ["k_fwd_bwd_mid", "k_chain_wide", "k_dw_wide", "k_reduce_wide"],
        "mn": ["k_fwd_bwd_tall"]}
PASS = {w: PASS[w] for w in os.environ.get("PROFILE_WORKLOADS", "c2 c4 c5 mn").split() if w in PASS}


def kname(full):
    for k in ("k_fwd_bwd_fast3", "k_fwd_bwd_fast", "k_fwd_bwd_mid", "k_fwd_bwd_tall", "k_chain_wide", "k_dw_wide", "k_reduce_wide", "k_update", "k_hyper", "k_energy"):
        if k in full:
            return k
    return None


kernel_us, traffic = {}, {}
for w, names in PASS.items():
    f = os.path.join(out, f"{w}_kernel_stats.csv")
    if os.path.exists(f):
        rows = list(csv.DictReader(open(f)))
        per = {}
        for r in rows:
            k = kname(r["Name"])
            if k in names and int(r["Calls"]) > 5:
                per[k] = per.get(k, 0.0) + float(r["AverageNs"]) / 1e3
        if per:
            kernel_us[w] = {"us": round(sum(per.values()), 3), "kernels": {k: round(v, 3) for k, v in per.items()},
                            "source": f"profiles/{tag}_{w}_kernel_stats.csv"}
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, f"pmc_{w}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k in names:
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    if acc:
        summ = {}
        for (k, c), (v, n) in sorted(acc.items()):
            summ.setdefault(k, {})[c] = {"mean_per_launch": v / n, "launches": n}
        json.dump(summ, open(os.path.join(out, f"{w}_pmc_summary.json"), "w"), indent=1)
        rd = sum(2.0 * 1024 * summ[k]["FETCH_SIZE"]["mean_per_launch"] for k in summ if "FETCH_SIZE" in summ[k])
        wr = sum(1024.0 * summ[k]["WRITE_SIZE"]["mean_per_launch"] for k in summ if "WRITE_SIZE" in summ[k])
        if rd or wr:
            traffic[w] = {"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
                          "kernels": " + ".join(sorted(summ)),
                          "correction": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM)",
                          "source": f"profiles/{tag}_{w}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"}
# what was profiled: the loaded library's own id (hash of its sources, tbnn_build_id), its file hash, the commit it was built at
# (TBNN_GIT_HEAD: the GPU box has no .git) -- bench.py quotes these files only on a build_id match
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib
from tensorbnn_amd import _native as nat
build = {"build_id": nat.build_id(), "lib_sha256": hashlib.sha256(open(nat.LIB_PATH, "rb").read()).hexdigest(),
         "git_head": os.environ.get("TBNN_GIT_HEAD") or None, "tag": tag}
kernel_us["_build"] = build
traffic["_build"] = build
json.dump(kernel_us, open(os.path.join(out, "rocprof_kernel_us.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(kernel_us, indent=1)); print(json.dumps(traffic, indent=1))
