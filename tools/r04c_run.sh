#!/bin/bash
# A/B of one spg_tune switch on one box + selected tests + kernel trace:   tools/r04c_run.sh <tag> "<tuneA>" "<tuneB>" [pytest args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r04c}; TA="$2"; TB="$3"; shift 3
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest -m gpu -q -x "$@" 2>&1 | tail -30 > $OUT/${TAG}_pytest.txt
cat $OUT/${TAG}_pytest.txt
{
  AB_ARGS="" bash tools/ab.sh "$TA" "$TB" 3
  echo "2 scenes / step"; AB_ARGS="--scenes 2" bash tools/ab.sh "$TA" "$TB" 2
  echo "8 scenes / step"; AB_ARGS="--scenes 8" bash tools/ab.sh "$TA" "$TB" 1
} 2>&1 | tee $OUT/${TAG}_ab.txt
bash tools/quick_trace.sh $TAG
