#!/bin/bash
# dev: A/B of C2 kernel build variants: ./tools/c2_variants.sh "<flags1>" "<flags2>" ...
for V in "$@"; do
  echo "=== variant: [$V]"
  TBNN_EXTRA_FLAGS="$V" python3 -m tensorbnn_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  for r in 1 2; do timeout 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   steps/s', d['value'], 'kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'], 'acc', d['accept_ratio'])"; done
done
python3 -m tensorbnn_amd.build --force > /dev/null 2>&1
