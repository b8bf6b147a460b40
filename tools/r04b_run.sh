#!/bin/bash
# round-4 re-entry GPU call: the GPU tests, A/B of the fused backward (spg_tune key 14 = 1 restores the separate launches),
# kernel trace summary + launch sequence of the last step     tools/r04b_run.sh <tag> [pytest args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r04b}; shift
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 > $OUT/${TAG}_pytest.txt
cat $OUT/${TAG}_pytest.txt
AB_ARGS="" bash tools/ab.sh "14:1" "" 3 2>&1 | tee $OUT/${TAG}_ab_bwdpair.txt
bash tools/quick_trace.sh $TAG
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-live-pmc > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
head -c 600 $OUT/${TAG}_bench.json
