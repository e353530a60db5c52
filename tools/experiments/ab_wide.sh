#!/bin/bash
# dev: A/B the wide kernels under different TBNN_WIDE_FLAGS: tools/ab_wide.sh <c4|c5> "<flags A>" "<flags B>" ...
wl=$1; shift
for fl in "$@"; do
  TBNN_WIDE_FLAGS="$fl" python3 -c "from tensorbnn_amd import build as b; b.build(force=True, verbose=False)" 2>&1 | grep -E "error" | head -3
  r=$(python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"kernel_us": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')
  echo "[$wl][$fl] $r"
done
