#!/bin/bash
# Copy one tools/profile_round.sh result set from gpurun_out/<tag>/ into profiles/ (tracked): the kernel-stat CSVs, the PMC
# summaries, the bench line each profiled command printed (the line the CSV's durations are to be held against: same run, same
# state), and the two JSON pointer files bench.py reads (they carry the build id of the library that was profiled).
#   tools/collect_profiles.sh r04a
T=$1
S=gpurun_out/$T
for W in c2 c4 c5 mn w300 mc10 wm10; do
  [ -f $S/${W}_kernel_stats.csv ] && cp $S/${W}_kernel_stats.csv profiles/${T}_${W}_kernel_stats.csv
  [ -f $S/${W}_pmc_summary.json ] && cp $S/${W}_pmc_summary.json profiles/${T}_${W}_pmc_summary.json
  [ -f $S/trace_$W.log ] && grep '^{"metric"' $S/trace_$W.log | tail -1 > profiles/${T}_${W}_bench_line_under_rocprof.json
done
cp $S/rocprof_kernel_us.json $S/pmc_traffic.json profiles/
cp $S/rocprof_kernel_us.json profiles/${T}_rocprof_kernel_us.json; cp $S/pmc_traffic.json profiles/${T}_pmc_traffic.json
ls profiles | grep $T
