"""dev: print the top kernels of rocprofv3 kernel_stats.csv files:  python tools/experiments/kstats.py gpurun_out/tv_*"""
import csv, sys, os
for d in sys.argv[1:]:
    f = d if d.endswith(".csv") else os.path.join(d, "kernel_stats.csv")
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 10]
    print(d, " | ".join(f"{r['Name'].split('(')[0].split('<')[0].replace('void ', '')}{'<' + r['Name'].split('<', 1)[1][:14] if '<' in r['Name'] and 'lay' in r['Name'] else ''} x{r['Calls']} {float(r['AverageNs']) / 1e3:.2f}us" for r in rows[:7]))
