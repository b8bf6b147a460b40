#!/bin/bash
# A/B of two BUILDS of libspg_hip.so on ONE box, interleaved:  tools/ab_builds.sh <libB.so> [reps] [bench args...]
#   A = the in-tree library, B = the given file (SPG_HIP_LIB, superpoint_graph_amd/_lib.py).  Boxes differ by +-3 %: never compare
#   numbers of different gpurun calls.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B="$1"; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for L in "" "$B"; do
    SPG_HIP_LIB="$L" python $ROOT/bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=[${L:-in-tree}]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4))"
  done
done
