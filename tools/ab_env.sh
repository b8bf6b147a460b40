#!/bin/bash
# A/B of an environment variable on ONE box, interleaved:  tools/ab_env.sh NAME "<valueA>" "<valueB>" [reps]   ("" = unset)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME="$1"; A="$2"; B="$3"; N=${4:-3}
for i in $(seq $N); do
  for V in "$A" "$B"; do
    if [ -z "$V" ]; then unset $NAME; else export $NAME="$V"; fi
    python $ROOT/bench.py $AB_ARGS --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$NAME=[$V]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4))"
  done
done
