#!/bin/bash
# Semantic3D-scale A/B of spg_tune switches on one box:  tools/ab_sema3d.sh "<tuneA>" "<tuneB>" [reps] [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
A="$1"; B="$2"; N=${3:-2}; shift; shift; shift
for i in $(seq $N); do
  for T in "$A" "$B"; do
    python $ROOT/bench.py --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 --steps 10 --warmup 3 --tune "$T" "$@" --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$T]', round(d['ms_per_step'],4), round(d['value']), d['self_check'] if 'self_check' in d else '')"
  done
done
