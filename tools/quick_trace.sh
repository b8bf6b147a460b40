#!/bin/bash
# kernel trace of a few steady-state steps -> gpurun_out/<tag>_kernel_stats.txt + <tag>_timeline.txt   (tools/quick_trace.sh <tag> [bench args])
TAG=${1:-quick}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_q_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_q_$TAG -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras "$@" > /dev/null 2> $OUT/${TAG}_trace.err
DB=$(find /tmp/p_q_$TAG -name '*.db' | head -1)
python $ROOT/tools/prof_summary.py $DB 60 > $OUT/${TAG}_kernel_stats.txt
python $ROOT/tools/prof_timeline.py $DB 25 > $OUT/${TAG}_timeline.txt
head -3 $OUT/${TAG}_kernel_stats.txt | cut -c1-200
