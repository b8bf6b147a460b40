#!/bin/bash
# round-4 GPU call: tests, A/B (module path ungrouped / module path grouped / fused step), kernel trace summary + the launch
# sequence of the last step   tools/r04_run.sh <tag> [pytest args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r04x}; shift
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 > $OUT/${TAG}_pytest.txt
cat $OUT/${TAG}_pytest.txt
B="--steps 40 --warmup 10 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras"
for i in 1 2; do
  for V in "--fused-step 0 --tune 11:1" "--fused-step 0" "--fused-step 1"; do
    python bench.py $B $V 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$V]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4))"
  done
done 2>&1 | tee $OUT/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_trace
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras > /dev/null 2> $OUT/${TAG}_trace.err
DB=$(find /tmp/p_trace -name '*.db' | head -1)
python $ROOT/tools/prof_summary.py $DB 70 > $OUT/${TAG}_kernel_stats.txt
python $ROOT/tools/prof_timeline.py $DB 25 > $OUT/${TAG}_timeline.txt
head -8 $OUT/${TAG}_kernel_stats.txt
