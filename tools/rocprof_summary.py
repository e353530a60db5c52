"""Summaries of one tools/profile_round.sh run: per-workload PMC means per launch of the fused-pass kernels, the kernel time
of one fused pass from the rocprofv3 kernel trace (rocprof_kernel_us.json: what bench.py's roofline.frac_rocprof uses) and the
HBM traffic per pass with the gfx950 correction (pmc_traffic.json: roofline.traffic).  python tools/rocprof_summary.py <dir> [tag]

Kernel time of a pass = sum over its kernels of the MEDIAN launch duration from the per-dispatch trace (<w>_kernel_trace.csv is
read before profile_round.sh deletes it); next to it the mean, the 1 %-trimmed mean, the maximum and the count of launches
longer than 3 x the median (round 4: ONE 31.7-ms launch of k_chain_wide among 1,210 moved its mean by 0.9 %, which is more than
the gap between the tracked summary and the driver's line).  Without a trace the stats CSV's AverageNs is used and marked so."""
import collections, csv, glob, json, os, sys

import numpy as np

out = sys.argv[1]
tag = os.path.basename(os.path.normpath(out))
# kernels of one fused forward+backward pass, per workload
PASS = {"c2": ["k_fwd_bwd_fast3"], "c4": ["k_chain_wide", "k_dw_wide", "k_reduce_wide"], "c5": ["k_fwd_bwd_mid", "k_chain_wide", "k_dw_wide", "k_reduce_wide"],
        "mn": ["k_fwd_bwd_tall"],
        # the layered family (any architecture): every launch of one gradient, counted per leapfrog step through launches / passes
        "w300": ["k_lay_"], "mc10": ["k_lay_", "k_fwd_bwd_tall", "k_fwd_bwd_mid"], "wm10": ["k_chain_wide", "k_dw_wide", "k_reduce_wide", "k_lay_"],
        "oh100": ["k_fwd_bwd_fast3", "k_fwd_bwd_fast", "k_lay_"], "wf50": ["k_chain_wide", "k_dw_wide", "k_reduce_wide", "k_lay_"]}
PASS = {w: PASS[w] for w in os.environ.get("PROFILE_WORKLOADS", "c2 c4 c5 mn w300 mc10 wm10").split() if w in PASS}
KNOWN = ("k_fwd_bwd_fast3", "k_fwd_bwd_fast", "k_fwd_bwd_mid", "k_fwd_bwd_tall", "k_chain_wide", "k_dw_wide", "k_reduce_wide", "k_update", "k_hyper",
         "k_energy", "k_begin")


def kname(full):
    if "k_lay_" in full:                      # layered family: k_lay_gemm<...>, k_lay_dw, k_lay_tail, k_lay_lik ... keep the template-free name
        return full[full.index("k_lay_"):].split("<")[0].split("(")[0]
    for k in KNOWN:
        if k in full:
            return k
    return None


def trace_durations(w):
    """kernel -> array of launch durations (ns) from the per-dispatch trace of workload w, or {}"""
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, f"trace_{w}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                per[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: np.asarray(v, dtype=np.float64) for k, v in per.items()}


def robust(d):
    d = np.sort(d)
    cut = int(len(d) * 0.01)
    tm = d[cut:len(d) - cut] if len(d) > 2 * cut + 1 else d
    med = float(np.median(d))
    return {"median_us": round(med / 1e3, 3), "mean_us": round(float(d.mean()) / 1e3, 3), "trimmed_mean_us": round(float(tm.mean()) / 1e3, 3),
            "min_us": round(float(d[0]) / 1e3, 3), "max_us": round(float(d[-1]) / 1e3, 3), "launches": int(len(d)),
            "launches_over_3x_median": int((d > 3.0 * med).sum())}


kernel_us, traffic = {}, {}
for w, names in PASS.items():
    dur = trace_durations(w)
    sel = {k: v for k, v in dur.items() if any(k.startswith(nm) for nm in names) and len(v) > 5}
    if sel:
        stats = {k: robust(v) for k, v in sel.items()}
        # launches of one pass: the kernels of the c2 / c4 / c5 / mn passes run once per pass; a layered gradient launches some
        # kernels several times (one GEMM per layer), so weight every kernel by launches / passes (passes = the rarest kernel's count)
        passes = min(s["launches"] for s in stats.values())
        # a kernel launched once per pass: its median; one launched several times per pass with different shapes (the layered GEMMs:
        # a median over unlike launches means nothing): 1 %-trimmed mean x launches per pass
        per_pass = lambda s: s["median_us"] if s["launches"] == passes else s["trimmed_mean_us"] * s["launches"] / passes
        us = sum(per_pass(s) for s in stats.values())
        kernel_us[w] = {"us": round(us, 3), "statistic": "sum over the pass's kernels of the median launch duration (kernels launched several times per pass: 1 %-trimmed mean x launches per pass)",
                        "mean_us": round(sum(s["mean_us"] * s["launches"] / passes for s in stats.values()), 3),
                        "kernels": {k: round(per_pass(s), 3) for k, s in stats.items()}, "detail": stats, "passes": passes,
                        "others": {k: robust(v) for k, v in dur.items() if k not in sel and len(v) > 5},
                        "source": f"profiles/{tag}_{w}_kernel_stats.csv + per-dispatch trace (medians: profiles/{tag}_rocprof_kernel_us.json)"}
    else:
        f = os.path.join(out, f"{w}_kernel_stats.csv")
        if os.path.exists(f):
            per = {}
            for r in csv.DictReader(open(f)):
                k = kname(r["Name"])
                if k and any(k.startswith(nm) for nm in names) and int(r["Calls"]) > 5:
                    per[k] = per.get(k, 0.0) + float(r["AverageNs"]) / 1e3
            if per:
                kernel_us[w] = {"us": round(sum(per.values()), 3), "statistic": "AverageNs of the stats CSV (no per-dispatch trace found)",
                                "kernels": {k: round(v, 3) for k, v in per.items()}, "source": f"profiles/{tag}_{w}_kernel_stats.csv"}
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, f"pmc_{w}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k and any(k.startswith(nm) for nm in names):
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    if acc:
        summ = {}
        for (k, c), (v, n) in sorted(acc.items()):
            if n > 5:                      # (one-off kernels -- the layered family packs the rows once per data set -- are not part of a pass)
                summ.setdefault(k, {})[c] = {"mean_per_launch": v / n, "launches": n}
        json.dump(summ, open(os.path.join(out, f"{w}_pmc_summary.json"), "w"), indent=1)
        # per PASS: a kernel launched several times per pass (layered GEMMs) counts launches / passes times
        rd = sum(2.0 * 1024 * summ[k]["FETCH_SIZE"]["mean_per_launch"] * summ[k]["FETCH_SIZE"]["launches"] for k in summ if "FETCH_SIZE" in summ[k])
        wr = sum(1024.0 * summ[k]["WRITE_SIZE"]["mean_per_launch"] * summ[k]["WRITE_SIZE"]["launches"] for k in summ if "WRITE_SIZE" in summ[k])
        prd = min([summ[k]["FETCH_SIZE"]["launches"] for k in summ if "FETCH_SIZE" in summ[k]] or [1])
        pwr = min([summ[k]["WRITE_SIZE"]["launches"] for k in summ if "WRITE_SIZE" in summ[k]] or [1])
        rd, wr = rd / prd, wr / pwr
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
brief = {w: {k: v for k, v in e.items() if k not in ("detail", "others")} for w, e in kernel_us.items()}
print(json.dumps(brief, indent=1)); print(json.dumps(traffic, indent=1))
