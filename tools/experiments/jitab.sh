#!/bin/bash
# dev: A/B of a run-time compiled kernel variant on the GPU box: alternating bench.py runs with and without TBNN_JIT_FLAGS="$1"
# (the variant libraries are compiled HERE first: TBNN_JIT_FLAGS=... python3 tools/experiments/jit_build.py 8,50,50,1)
#   bash tools/experiments/jitab.sh "-DTBNN_F3_DENSE=0" c2 200
FLAGS="$1"; W=${2:-c2}; STEPS=${3:-200}
for rep in 1 2 3; do
  for v in base var; do
    if [ $v = var ]; then export TBNN_JIT_FLAGS="$FLAGS"; else unset TBNN_JIT_FLAGS; fi
    timeout -k 10 300 python3 bench.py --workload $W --steps $STEPS --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('$v', d['value'], 'steps/s', r.get('kernel_us'), 'us')" || exit 1
  done
done
