"""dev: accept ratio of bench.py's exact schedules (driver: --steps 20 --warmup 5; default: the workload's own) from the
burned-in fixtures at candidate step sizes -- the chain RNG is keyed by (seed, chain_id, epoch), so these numbers repeat"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cands = {"c2": [7.6e-5, 8.0e-5, 8.4e-5, 8.8e-5], "c5": [5.0e-5, 6.0e-5, 7.0e-5, 8.3e-5], "c5g": [3.0e-4, 3.7e-4], "c4": [1.3e-5, 1.455e-5]}
for wl in sys.argv[1:] or ["c2", "c5"]:
    for eps in cands[wl]:
        row = []
        for extra in (["--steps", "20", "--warmup", "5"], []):
            if wl != "c2" and extra:
                continue
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--eps", str(eps)] + extra,
                               capture_output=True, text=True)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
            row.append((d["steps"], d["accept_ratio"], d.get("hyper_accept_ratio")))
        print(wl, eps, row, flush=True)
