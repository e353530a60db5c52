#!/bin/bash
# dev (GPU): one architecture on each family that can take it (TBNN_JIT_SKIP) and on the layered family: which is worth extending
#   tools/experiments/fam_time.sh 20,100,100,2 100000 bern
D=$1; N=$2; shift 2
for SK in "" "fast3,fast,mid" "fast3,fast,mid,tall,wide"; do
  echo "== TBNN_JIT_SKIP=$SK"
  TBNN_JIT_SKIP=$SK TBNN_MID=$([ -z "$SK" ] && echo 1 || echo 0) timeout -k 10 200 python tools/experiments/shape_time.py $D $N "$@" 2>&1 | grep -v Warning | tail -3
done
