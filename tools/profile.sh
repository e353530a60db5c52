#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace + stats, then separate PMC passes (never combined with traces)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
BENCH="python3 bench.py --workload c2 --steps 40 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/pmc_lds.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
python3 - <<PY
import csv, glob, collections, json, os
out="$OUT"
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh): yield r
summ={}
for tag in ("pmc_fetch","pmc_write","pmc_sq","pmc_lds"):
    acc=collections.defaultdict(lambda: [0.0,0])
    for r in rows(f"{tag}/**/*counter_collection.csv"):
        k=r.get("Kernel_Name","")
        if "k_fwd_bwd_fast" not in k or "Li5ELi50" not in k and "5, 50, 50, 50" not in k: continue
        acc[r["Counter_Name"]][0]+=float(r["Counter_Value"]); acc[r["Counter_Name"]][1]+=1
    for c,(v,n) in acc.items(): summ[c]={"mean_per_launch": v/max(n,1), "launches": n}
json.dump(summ, open(os.path.join(out,"pmc_summary_fwd_bwd_fast.json"),"w"), indent=1)
print(json.dumps(summ))
PY
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
cat $OUT/kernel_stats.csv | cut -c1-200
# keep only the small summaries (the per-dispatch CSVs are large)
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
