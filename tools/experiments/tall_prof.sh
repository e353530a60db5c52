#!/bin/bash
# dev: rocprofv3 kernel stats of one shape through shape_time.py:  tools/experiments/tall_prof.sh <tag> "<dims> <rows> [bern]"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
mkdir -p gpurun_out/$T
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/trace -- python3 tools/experiments/shape_time.py $@ > gpurun_out/$T/run.log 2>&1
tail -2 gpurun_out/$T/run.log
f=$(find gpurun_out/$T/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/$T/kernel_stats.csv && cut -d, -f1-4 $f | head -8
find gpurun_out/$T -name "*kernel_trace.csv" -delete; find gpurun_out/$T -name "*agent_info.csv" -delete
