"""dev: the launches of ONE gradient pass in order, with durations, from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --workload mc10 --eps 1e-6 --steps 10 --warmup 2 --no-cpu-baseline
  python tools/experiments/pass_trace.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "")[:70] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
grid = [(r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows]
# find the period: positions of k_update, take a window between two k_update launches late in the run
upd = [i for i, n in enumerate(names) if n.startswith("k_update")]
a, b = upd[-3], upd[-2]
med = collections.defaultdict(list)
for i in range(len(rows)):
    med[(names[i], grid[i])].append(dur[i])
for i in range(a + 1, b + 1):
    v = sorted(med[(names[i], grid[i])])
    gap = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
    print(f"{names[i]:72s} grid {int(grid[i][0]) // int(grid[i][1]):5d}  {dur[i]:8.2f} us (median of its kind {v[len(v) // 2]:8.2f})  gap before {gap:5.2f}")
print("pass total", sum(dur[a + 1:b + 1]), "wall", (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
