#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
AB_ARGS="--n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 --steps 10 --warmup 3" bash $ROOT/tools/ab_libs.sh "$@"
