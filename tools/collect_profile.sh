#!/bin/bash
# Copies what tools/profile_bench.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (the committed,
# judged copies) and refreshes profiles/rank2_traffic.json (bench.py's `roofline.traffic`).  Run in the authoring container.
set -e
TAG=${1:-r04}
SRC=gpurun_out/prof_$TAG
DST=profiles/$TAG
mkdir -p $DST
cp $SRC/summary.txt $DST/summary.txt
# (gpurun merges every call's files into gpurun_out/: take the newest run's)
cp "$(ls -t $SRC/trace/*/*_kernel_stats.csv | head -1)" $DST/kernel_stats.csv
cp $SRC/bench_trace.json $DST/bench_under_trace.json
# the PMC rows of the contract kernel only (the per-dispatch files hold every launch of every kernel)
for k in fetch write; do
  f=$(ls -t $SRC/pmc_$k/*/*_counter_collection.csv | head -1)
  { head -1 $f; grep "k_rank2<\|k_rank2_queue<" $f; } > $DST/pmc_${k}_size.csv
done
cp $SRC/rank2_traffic.json profiles/rank2_traffic.json
echo "collected into $DST; profiles/rank2_traffic.json refreshed"
