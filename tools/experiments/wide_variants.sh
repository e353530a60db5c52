#!/bin/bash
# dev: build diagnostic variants of the wide kernels on the GPU box and time them
for V in "-DWIDE_DW_PD=1" "-DWIDE_DW_PD=2" "-DWIDE_DW_PD=3 -DWIDE_PD=1" "-DWIDE_DW_PD=4"; do
  echo "=== variant: [$V]"
  TBNN_EXTRA_FLAGS="$V" python3 -m tensorbnn_amd.build --force > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/var; mkdir -p gpurun_out/var
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var -- python3 tools/widetime.py ${1:-c4} 6 > gpurun_out/var.log 2>&1
  f=$(find gpurun_out/var -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:2]:
    print("   ", r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3, "us")
PY
done
python3 -m tensorbnn_amd.build --force > /dev/null 2>&1
