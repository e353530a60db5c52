#!/bin/bash
# dev (GPU): WHICH LDS accesses of configs[1]'s kernel carry its bank-conflict cycles.  The skeleton builds of
# kernels_fast.hpp (TBNN_SKEL bits: 1 = no row tiles, 2 = no non-operand VALU work / fringe rows, 4 = the chains' weight operands not read from LDS,
# 8 = no fourth N tile of dW) under one PMC pass each: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS per launch of k_fwd_bwd_fast3.
# Differences between variants attribute the conflict cycles (wrong results by construction: diagnostic builds, never the product).
#   tools/experiments/lds_conflicts.sh        (results: gpurun_out/ldsc/summary.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ldsc; mkdir -p $OUT
rm -f $OUT/summary.txt
for V in ${SKEL_VARIANTS:-0 1 4 2 6 8}; do
  if [ $V = 0 ]; then LIB=tensorbnn_amd/libtbnn.so; else
    [ -f tensorbnn_amd/libtbnn_skel$V.so ] || TBNN_BUILD_TAG=skel$V TBNN_EXTRA_FLAGS=-DTBNN_SKEL=$V TBNN_ALLOW_SPILL=1 python3 -m tensorbnn_amd.build > $OUT/build_$V.log 2>&1 || { echo "build skel$V failed" >> $OUT/summary.txt; tail -5 $OUT/build_$V.log; continue; }
    LIB=tensorbnn_amd/libtbnn_skel$V.so
  fi
  export TBNN_LIB=$GRAFT_REPO_ROOT/$LIB
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_$V -- python3 bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --repeats 1 > $OUT/bench_$V.log 2>&1 || echo "run $V failed" >> $OUT/summary.txt
  unset TBNN_LIB
  python3 - $OUT/pmc_$V $V >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for path in f:
    for r in csv.DictReader(open(path)):
        if "k_fwd_bwd_fast3" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("TBNN_SKEL=%s  " % sys.argv[2] + "  ".join("%s %.4g" % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())))
PY
  find $OUT/pmc_$V -name "*.csv" -size +1M -delete
done
cat $OUT/summary.txt
