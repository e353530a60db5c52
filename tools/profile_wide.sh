#!/bin/bash
# rocprofv3 kernel trace of the wide path at full size: tools/profile_wide.sh <tag> <case>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r01w}; CASE=${2:-c4}
OUT=gpurun_out/$R
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CASE -- python3 tools/widetime.py $CASE 10 > $OUT/trace_$CASE.log 2>&1
tail -3 $OUT/trace_$CASE.log | cut -c1-300
f=$(find $OUT/trace_$CASE -name "*kernel_stats.csv" | head -1)
cp $f $OUT/${CASE}_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
