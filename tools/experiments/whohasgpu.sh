#!/bin/bash
# dev: which processes hold /dev/kfd while `bench.py --gpus 3` (one-GPU rehearsal hook) runs
cd $GRAFT_REPO_ROOT
hipcc -O2 -std=c++17 -shared -fPIC -o tests/stubccl/libstubccl.so tests/stubccl/stub_ccl.cpp -lrt 2>/dev/null
TBNN_BENCH_SINGLE_GPU=1 TBNN_RCCL_LIB=$PWD/tests/stubccl/libstubccl.so python bench.py --gpus 3 --steps 200 --warmup 2 --sampling-step 50 > gpurun_out/who_bench.log 2>&1 &
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  sleep 2
  echo "--- t=$((2*i))s"
  for p in /proc/[0-9]*; do
    if ls -l $p/fd 2>/dev/null | grep -q "kfd\|renderD"; then echo "$(basename $p) $(tr '\0' ' ' < $p/cmdline | cut -c1-110)"; fi
  done
done
wait
tail -2 gpurun_out/who_bench.log | cut -c1-300
