#!/bin/bash
# One short GPU-box call: the GPU test suite, the default bench line and the steady-state kernel statistics.
#   tools/quick_check.sh <tag> [pytest args...]   ->  gpurun_out/<tag>_{tests.log,bench.json,kernel_stats.txt}
set -u
TAG=${1:-chk}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x "$@" > $OUT/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log
timeout 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
STEPS="--no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace -- python $ROOT/bench.py --steps 20 --warmup 5 $STEPS > /dev/null 2> $OUT/${TAG}_trace.err
python $ROOT/tools/prof_summary.py $(find /tmp/p_trace -name '*.db' | head -1) 70 > $OUT/${TAG}_kernel_stats.txt
head -12 $OUT/${TAG}_kernel_stats.txt
