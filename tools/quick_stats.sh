#!/bin/bash
# kernel statistics of the steady-state training step only:  tools/quick_stats.sh <tag> [bench flags]  ->  gpurun_out/<tag>_kernel_stats.txt
set -u
TAG=${1:-chk}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_trace_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace_$TAG -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras "$@" > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
python $ROOT/tools/prof_summary.py $(find /tmp/p_trace_$TAG -name '*.db' | head -1) 80 > $OUT/${TAG}_kernel_stats.txt
head -4 $OUT/${TAG}_kernel_stats.txt; grep -E "ecc|copy2d" $OUT/${TAG}_kernel_stats.txt
