#!/bin/bash
# dev: k_fwd_bwd_tall by group size (TBNN_TALL_G) and row count, rocprofv3 kernel time:  tools/experiments/tall_gsweep.sh "<dims>" [bern] -- n1 n2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=$1; shift; L=""; if [ "$1" != "--" ]; then L=$1; shift; fi; shift
for n in $@; do
  line="n=$n"
  for g in 1 2 3 4 auto; do
    rm -rf gpurun_out/gs; mkdir -p gpurun_out/gs
    if [ $g = auto ]; then unset TBNN_TALL_G; else export TBNN_TALL_G=$g; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gs/trace -- python3 tools/experiments/shape_time.py $D $n $L > gpurun_out/gs/run.log 2>&1
    f=$(find gpurun_out/gs/trace -name "*kernel_stats.csv" | head -1)
    us=$(grep k_fwd_bwd_tall $f | awk -F, '{print $(NF-4)/1000}' | head -1)
    line="$line | G=$g $us"
  done
  echo "$line"
done
