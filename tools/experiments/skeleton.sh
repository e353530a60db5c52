#!/bin/bash
# dev / DESIGN 4.1: what the one-wave-per-SIMD design of k_fwd_bwd_fast3 can reach at configs[1].  Builds three DIAGNOSTIC variants of the
# library next to the product (TBNN_BUILD_TAG; wrong results by construction, never shipped) and takes the rocprofv3 duration of the
# fused launch under the bench command for each:
#   skel1: no row tiles (prologue + epilogue only)            -> the launch's fixed cost
#   skel2: tile body without its non-operand VALU work        -> MFMAs + the LDS traffic their operands need
#   skel3: both                                                -> (control: equals skel1)
# plus the MFMA-only stream of a tile in registers (tools/ubench/skel_c2.hip): the floor of the tile loop at the clock the chip holds.
#   tools/experiments/skeleton.sh      (on the GPU box; results: gpurun_out/skel/summary.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/skel; mkdir -p $OUT
rm -f $OUT/summary.txt
for V in ${SKEL_VARIANTS:-0 1 2 4 8 14}; do
  if [ $V = 0 ]; then LIB=tensorbnn_amd/libtbnn.so; else
    TBNN_BUILD_TAG=skel$V TBNN_EXTRA_FLAGS=-DTBNN_SKEL=$V TBNN_ALLOW_SPILL=1 python3 -m tensorbnn_amd.build > $OUT/build_$V.log 2>&1 || { echo "build skel$V failed"; tail -5 $OUT/build_$V.log; continue; }
    LIB=tensorbnn_amd/libtbnn_skel$V.so
  fi
  export TBNN_LIB=$GRAFT_REPO_ROOT/$LIB
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$V -- python3 bench.py --workload c2 --steps 40 --warmup 10 --no-cpu-baseline --repeats 1 > $OUT/bench_$V.log 2>&1 || echo "run $V failed"
  f=$(find $OUT/trace_$V -name "*kernel_stats.csv" | head -1)
  echo "== TBNN_SKEL=$V" >> $OUT/summary.txt
  [ -n "$f" ] && python3 - $f >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_fwd_bwd_fast3" in r["Name"] or r["Name"].startswith("k_update"):
        print("   %-18s calls %5s  mean %8.2f us  min %7.2f  max %7.2f" % (r["Name"].split("<")[0].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  grep -o '"value": [0-9.]*' $OUT/bench_$V.log | head -1 >> $OUT/summary.txt
  TBNN_LIB=$GRAFT_REPO_ROOT/$LIB timeout -k 10 120 python3 tools/stamps.py 2>/dev/null | tail -4 >> $OUT/summary.txt
  unset TBNN_LIB
  find $OUT/trace_$V -name "*kernel_trace.csv" -delete
done
if [ -f tools/ubench/skel_c2.hip ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $OUT/skel_c2 tools/ubench/skel_c2.hip && timeout -k 10 120 $OUT/skel_c2 >> $OUT/summary.txt 2>&1
fi
cat $OUT/summary.txt
