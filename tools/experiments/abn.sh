#!/bin/bash
# dev: alternate bench runs of the product library and several diagnostic builds (TBNN_BUILD_TAG=<tag> TBNN_EXTRA_FLAGS=... python -m tensorbnn_amd.build):
#   tools/experiments/abn.sh "<tag> <tag> ..." <workload> [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAGS=$1; W=${2:-c2}; K=${3:-200}
for rep in 1 2 3; do
  for T in product $TAGS; do
    if [ $T = product ]; then lib=""; else lib=$PWD/tensorbnn_amd/libtbnn_$T.so; fi
    echo -n "$T "
    TBNN_LIB=$lib timeout -k 10 300 python3 bench.py --workload $W --steps $K --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print(d['value'], 'fused us', r.get('kernel_us'), 'regions', d.get('timed_regions_ms'))"
  done
done
