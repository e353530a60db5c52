#!/bin/bash
# dev: k_lay_dw by number of row ranges (TBNN_LAY_DW_WAVES = NS x 4 NY) at 8-300-300-1, 5e4 rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in $@; do
  O=gpurun_out/lns_$W; mkdir -p $O
  TBNN_LAY_DW_WAVES=$W TBNN_JIT=0 TBNN_TALL=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/experiments/shape_time.py 8,300,300,1 50000 > $O/run.log 2>&1
  f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
  rm -rf $O/trace
  echo "W=$W $(python3 tools/experiments/kstats.py $O/kernel_stats.csv | cut -c1-400)"
  grep "us per leapfrog" $O/run.log
done
