#!/bin/bash
# dev: kernel stats + two PMC passes of one shape through shape_time.py:  tools/experiments/shape_pmc.sh <tag> <dims> <rows> [bern]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/experiments/shape_time.py $@ > $O/run.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$i -- python3 tools/experiments/shape_time.py $@ > $O/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
ks = sorted({k for k, _ in acc})
import json
out = {k: {c: round(v / n, 1) for (kk, c), (v, n) in acc.items() if kk == k} for k in ks}
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -1 $O/run.log
