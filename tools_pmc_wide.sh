#!/bin/bash
# dev: PMC counters of the wide kernels (separate passes, no trace domains)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmcw}; CASE=${2:-c4}
mkdir -p $OUT
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools_widetime.py $CASE 3 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
acc=collections.defaultdict(lambda: [0.0,0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        kk="chain" if "k_chain_wide" in k else "dw" if "k_dw_wide" in k else None
        if kk is None: continue
        a=acc[(kk, r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for (kk,c),(v,n) in sorted(acc.items()): print(kk, c, v/n, n)
PY
