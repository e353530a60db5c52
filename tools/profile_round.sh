#!/bin/bash
# One round's rocprofv3 evidence for profiles/: kernel trace + stats and separate PMC passes (never combined with a trace
# domain) of THE BENCH COMMAND ITSELF for BASELINE configs[1], configs[3] and configs[4] (bench.py --workload c2 | c4 | c5: the
# burned-in state and step size the driver's line is measured at, so that the dominant kernel's mean duration here can be
# held against ms_per_step / L of that line; rounds 1-3 profiled tools/widetime.py at the initial state for c4 / c5).
#   tools/profile_round.sh r03b        -> gpurun_out/r03b/{c2,c4,c5}_kernel_stats.csv, *_pmc_summary.json,
#                                         rocprof_kernel_us.json, pmc_traffic.json   (copy the ones to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
declare -A CMD
CMD[c2]="python3 bench.py --workload c2 --steps 40 --warmup 10 --no-cpu-baseline"
CMD[c4]="python3 bench.py --workload c4 --no-cpu-baseline"
CMD[c5]="python3 bench.py --workload c5 --no-cpu-baseline"
CMD[mn]="python3 bench.py --workload mn --no-cpu-baseline"
# the layered family (any architecture; no burned-in fixture: a small fixed step from the initial state -- the kernels' durations do not depend on the state)
CMD[w300]="python3 bench.py --workload w300 --eps 1e-6 --no-cpu-baseline"
CMD[mc10]="python3 bench.py --workload mc10 --eps 1e-6 --no-cpu-baseline"
CMD[wm10]="python3 bench.py --workload wm10 --eps 1e-6 --no-cpu-baseline"
CMD[oh100]="python3 bench.py --workload oh100 --eps 1e-6 --no-cpu-baseline"
CMD[wf50]="python3 bench.py --workload wf50 --eps 1e-6 --no-cpu-baseline"
for W in ${PROFILE_WORKLOADS:-c2 c4 c5 mn w300 mc10 wm10}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- ${CMD[$W]} > $OUT/trace_$W.log 2>&1 || echo "trace $W failed"
  f=$(find $OUT/trace_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${W}_kernel_stats.csv
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_${W}_$i -- ${CMD[$W]} > $OUT/pmc_${W}_$i.log 2>&1 || echo "pmc pass $i ($SET) of $W failed"
  done
done
python3 tools/rocprof_summary.py $OUT          # (reads the per-dispatch traces: medians)
# keep only the small summaries (the per-dispatch CSVs are large)
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT | head -40
