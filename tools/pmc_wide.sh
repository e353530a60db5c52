#!/bin/bash
# PMC counters of the wide kernels: separate passes, no trace domains, every pass under its own timeout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmcw}; CASE=${2:-c4}
mkdir -p $OUT
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $SET --output-format csv -d $OUT/${CASE}_p$i -- python3 tools/widetime.py $CASE 2 > $OUT/${CASE}_p$i.log 2>&1 || echo "pass $i ($SET) failed/timed out"
done
python3 - <<PY
import csv, glob, collections, os, json
acc=collections.defaultdict(lambda: [0.0,0])
for f in glob.glob("$OUT/${CASE}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        kk="k_chain_wide" if "k_chain_wide" in k else "k_dw_wide" if "k_dw_wide" in k else "k_reduce_wide" if "k_reduce_wide" in k else None
        if kk is None: continue
        a=acc[(kk, r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
summ={}
for (kk,c),(v,n) in sorted(acc.items()):
    summ.setdefault(kk,{})[c]={"mean_per_launch": v/n, "launches": n}
json.dump(summ, open("$OUT/${CASE}_pmc_summary.json","w"), indent=1)
print(json.dumps(summ))
PY
