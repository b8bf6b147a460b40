#!/bin/bash
# A/B/C... of spg_tune settings on ONE box, interleaved:  tools/tune_list.sh <reps> "<tune1>" "<tune2>" ...   (bench args in $AB_ARGS)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
for i in $(seq $N); do
  for T in "$@"; do
    python $ROOT/bench.py $AB_ARGS --tune "$T" --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$T]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4))"
  done
done
