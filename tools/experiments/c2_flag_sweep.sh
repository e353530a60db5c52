#!/bin/bash
# round 2: LLVM scheduling flags on the narrow translation unit, configs[1] fused-pass time (bench.py hipEvent pairs)
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env TBNN_EXTRA_FLAGS="$2" python -m tensorbnn_amd.build --force > /tmp/b.log 2>&1 || { echo build failed; tail -3 /tmp/b.log; return; }
  python bench.py --workload c2 --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us'])"; }
run "default" ""
run "max-ilp" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy"
run "igrouplp exact" "-mllvm -amdgpu-igrouplp-exact-solver"
run "no unclustered resched" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule"
run "O2" "-O2"
run "misched bottomup" "-mllvm -misched-bottomup"
run "misched topdown" "-mllvm -misched-topdown"
run "no post-RA sched" "-mllvm -enable-post-misched=false"
run "sched model off cluster" "-mllvm -amdgpu-disable-power-sched=true"
run "early-ifcvt off + no licm hoist" "-mllvm -disable-licm-promotion"
