#!/bin/bash
# dev (GPU): one architecture on the fused family the JIT picks, on the wide family, and on the layered family: what a family is worth for a shape
#   tools/experiments/fam_time.sh 20,100,100,5 100000 gauss
D=$1; N=$2; shift 2
for SK in "" "fast3,fast,mid,tall" "fast3,fast,mid,tall,wide"; do
  echo "== TBNN_JIT_SKIP=$SK"
  TBNN_JIT_SKIP=$SK timeout -k 10 200 python tools/experiments/shape_time.py $D $N "$@" 2>&1 | grep -v "Warning\|nat.Chain" | tail -3
done
