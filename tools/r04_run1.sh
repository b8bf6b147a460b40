#!/bin/bash
# round-4 GPU call 1: tests, A/B of the grouped launches (spg_tune key 11), kernel trace summary
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/r04a_pytest.txt
cat $OUT/r04a_pytest.txt
bash tools/ab.sh "11:1" "" 2 2>&1 | tee $OUT/r04a_ab_group.txt
cd /tmp && export TMPDIR=/tmp
STEPS="--no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline"
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace -- python $ROOT/bench.py --steps 20 --warmup 5 $STEPS > /dev/null 2> $OUT/r04a_trace.err
python $ROOT/tools/prof_summary.py $(find /tmp/p_trace -name '*.db' | head -1) 70 > $OUT/r04a_kernel_stats.txt
head -40 $OUT/r04a_kernel_stats.txt
