#!/bin/bash
# tools/bitcheck_ab.sh <variant>: fingerprints of the in-tree library and of superpoint_graph_amd/csrc/variants/libspg_<variant>.so on
# the bench workloads (unit scene, 2 scenes, Semantic3D shape, per-iteration ECC / separate backward launches); prints DIFF lines if any
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=$ROOT/superpoint_graph_amd/csrc/variants/libspg_$1.so
run() { python $ROOT/tools/bitcheck.py "$@" 2>/dev/null | sed 's/^[^ ]* //'; }
for cfg in "" "--scenes 2" "--n-sp 3000 --n-edges 15000 --n-feat 11 --model-config gru_10,f_8" "--tune 14:1" "--tune 17:1,18:1,22:1" "--tune 8:1,11:1"; do
  a=$(run $cfg); b=$(SPG_HIP_LIB=$V run $cfg)
  if [ "$a" = "$b" ] && [ -n "$a" ]; then echo "SAME [$cfg] $a"; else echo "DIFF [$cfg]"; echo "  in-tree: $a"; echo "  $1: $b"; fi
done
