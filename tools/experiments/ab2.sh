#!/bin/bash
# dev: A/B the product build under different TBNN_EXTRA_FLAGS: tools/ab2.sh "<flags A>" "<flags B>" ...
for fl in "$@"; do
  TBNN_EXTRA_FLAGS="$fl" python3 -c "from tensorbnn_amd import build as b; b.build(force=True, verbose=False)" 2>&1 | grep -E "error" | head -3
  r=$(python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"kernel_us": [0-9.]*' | tr '\n' ' ')
  echo "[$fl] $r"
done
