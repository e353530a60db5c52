#!/bin/bash
# dev: process census (every 0.5 s: who holds /dev/kfd) while the multi-process tests of tests/test_gpu_api.py run
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 120); do
    sleep 0.5
    n=0; l=""
    for p in /proc/[0-9]*; do
      if ls -l $p/fd 2>/dev/null | grep -q "kfd"; then n=$((n+1)); l="$l | $(basename $p):$(tr '\0' ' ' < $p/cmdline | cut -c1-70)"; fi
    done
    echo "t=$i n=$n $l"
  done ) > gpurun_out/census.log 2>&1 &
CENSUS=$!
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -q -m gpu -k "bench" > gpurun_out/t5.log 2>&1
kill $CENSUS 2>/dev/null
tail -3 gpurun_out/t5.log
awk '{print $2}' gpurun_out/census.log | sort | uniq -c
grep -m3 "n=[6-9]\|n=1[0-9]" gpurun_out/census.log | cut -c1-900
