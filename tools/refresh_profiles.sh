#!/bin/bash
# Runs HERE after tools/gpu_profile.sh has run on the GPU box for a workload: copies the small summaries of
# gpurun_out/prof_<tag>_<workload> into profiles/<tag>_<workload>_fast.
# Usage: tools/refresh_profiles.sh <tag> <workload> [...]
TAG=$1; shift
for WL in "$@"; do
  SRC=gpurun_out/prof_${TAG}_$WL
  DST=profiles/${TAG}_${WL}_fast
  mkdir -p $DST
  cp $SRC/bench.json $DST/bench.json
  cp $SRC/summary.json $DST/pmc_summary.json
  cp $SRC/summary.txt $DST/pmc_summary.txt
  cp $(ls -t $(find $SRC/trace -name "*kernel_stats.csv") | head -n 1) $DST/kernel_stats.csv   # (the newest: gpurun merges every run's files into the same directory)
  cp $(ls -t $(find $SRC/trace_solo -name "*kernel_stats.csv") | head -n 1) $DST/kernel_stats_solo.csv
  echo "$DST: $(ls $DST | tr '\n' ' ')"
done
# the counter figures bench.py reads (each gpu_profile.sh run merges its workload into the box's copy; the copy
# of the run that came last holds them all)
LAST=$(ls -t gpurun_out/prof_${TAG}_*/pmc_traffic_merged.json | head -n 1)
cp $LAST profiles/pmc_traffic.json && echo "profiles/pmc_traffic.json <- $LAST"
