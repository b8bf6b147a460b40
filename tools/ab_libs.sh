#!/bin/bash
# interleaved A/B of several BUILDS of libspg_hip.so on one box:  AB_ARGS="--scenes 8" tools/ab_libs.sh <reps> <lib1.so> <lib2.so> ...   ("-" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset SPG_HIP_LIB; else export SPG_HIP_LIB="$ROOT/$L"; fi
    python $ROOT/bench.py $AB_ARGS --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[$L]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4), round(d['value']))"
  done
done
