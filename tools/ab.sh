#!/bin/bash
# A/B of spg_tune switches on ONE box, interleaved:  tools/ab.sh "<tuneA>" "<tuneB>" [reps]   (e.g. tools/ab.sh "" "9:1" 3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do
  for T in "$A" "$B"; do
    python $ROOT/bench.py $AB_ARGS --tune "$T" --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$T]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4))"
  done
done
