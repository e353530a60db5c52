#!/bin/bash
# dev (GPU): k_dw_wide's workgroup share of the one-tile last layer (WIDE_LL_COST8, kernels_wide.hpp: dw_cost) -- variants compiled HERE first with
# TBNN_JIT_FLAGS=-DWIDE_LL_COST8=n; per variant the step time and the rocprofv3 durations of the two kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/llcost
export TBNN_JIT_SKIP=fast3,fast,mid,tall
for D in "10,200,200,10 100000 bern" "20,100,100,5 100000 gauss"; do
  for F in "" "-DWIDE_LL_COST8=2" "-DWIDE_LL_COST8=1" "-DWIDE_LL_COST8=6" "-DWIDE_DW_G=4" "-DWIDE_DW_G=4 -DWIDE_LL_COST8=2"; do
    export TBNN_JIT_FLAGS="$F"
    O=gpurun_out/llcost/t_$(echo "$D$F" | tr -c 'a-zA-Z0-9' _)
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/experiments/shape_time.py $D > $O.log 2>&1
    echo "== $D [$F]: $(grep 'us per leapfrog' $O.log)"
    python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print("     %-16s %9.1f us" % (r["Name"].split("<")[0].replace("void ", ""), float(r["AverageNs"]) / 1e3))
PY
    find $O -name "*kernel_trace.csv" -delete
  done
done
