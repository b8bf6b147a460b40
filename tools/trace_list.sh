#!/bin/bash
# per-kernel durations of the train step under several spg_tune settings:  tools/trace_list.sh <tag> <regex> "<tune1>" "<tune2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; RE=$2; shift 2
n=0
for T in "$@"; do
  n=$((n+1))
  bash $ROOT/tools/quick_trace.sh ${TAG}_$n --tune "$T" > /dev/null 2>&1
  echo "tune=[$T] $(head -1 $ROOT/gpurun_out/${TAG}_${n}_kernel_stats.txt | sed 's/.*launches (//; s/ per step.*= / launches, /; s/;.*//')"
  grep -E "$RE" $ROOT/gpurun_out/${TAG}_${n}_kernel_stats.txt | cut -c1-150
done
