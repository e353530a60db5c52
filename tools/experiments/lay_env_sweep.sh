#!/bin/bash
# dev (GPU): the layered family's run-time knobs at one workload:  tools/experiments/lay_env_sweep.sh mc10 "TBNN_LAY_DW_WAVES=1024 TBNN_LAY_DW_WAVES=4096 ..."
W=${1:-mc10}; shift
for KV in "X=0" $@; do
  v=$(env $KV python3 bench.py --workload $W --eps 1e-6 --no-cpu-baseline --repeats 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$KV: $v"
done
