#!/bin/bash
# dev: A/B of one environment switch on the layered family:  tools/experiments/lay_ab.sh <VAR> <a> <b> ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1; shift
for rep in 1 2; do
for val in $@; do
  for sh in "8,300,300,1 50000" "784,100,100,10 12000" "40,300,300,3 20000" "100,50,50,1 100000" "64,512,512,4 30000" "20,128,128,128,2 200000"; do
    echo -n "$V=$val $sh: "
    env $V=$val TBNN_JIT=0 TBNN_TALL=0 timeout -k 10 120 python3 tools/experiments/shape_time.py $sh | tail -1
  done
done
done
