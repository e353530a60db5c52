#!/bin/bash
# dev: k_update geometry variants (libtbnn_u<cols>x<groups>.so, -DUPD_COLS / -DUPD_GROUPS) on a few shapes, rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" $(ls tensorbnn_amd/libtbnn_u*.so); do
  for sh in "784,20,20,1 12000 bern" "8,300,300,1 50000" "100,50,50,1 100000"; do
    T=ug; rm -rf gpurun_out/$T; mkdir -p gpurun_out/$T
    TBNN_LIB=${lib:+$PWD/$lib} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 tools/experiments/shape_time.py $sh > gpurun_out/$T/run.log 2>&1
    f=$(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1)
    echo "${lib:-product} | $sh | $(grep k_update $f | cut -d, -f1,4 | cut -c1-12,100-) | $(grep 'us per' gpurun_out/$T/run.log | cut -c1-30)"
  done
done
