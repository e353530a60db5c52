#!/bin/bash
# dev: alternate bench runs of the product library and a diagnostic build:  tools/experiments/ab_lib.sh <tag> <workload> [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; W=${2:-c2}; K=${3:-200}
for rep in 1 2 3; do
  for lib in "" $PWD/tensorbnn_amd/libtbnn_$T.so; do
    echo -n "lib=${lib:-product} "
    TBNN_LIB=$lib timeout -k 10 300 python3 bench.py --workload $W --steps $K --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print(d['value'], d['unit'], 'fused us', r.get('kernel_us'), 'frac', r.get('frac'))"
  done
done
