#!/bin/bash
# Everything the round's measurement deliverables need, in one GPU-box call (run from the repo root):
#   tools/collect_profiles.sh <tag>            e.g. r03  ->  gpurun_out/<tag>_{bench.json,kernel_stats.txt,pmc_*.txt,gemm_traffic.json}
# Passes: (1) bench.py (default flags: the driver's command), (2) rocprofv3 --kernel-trace of the train steps, summarised
# over the steady-state window (tools/prof_summary.py), (3)+(4) --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes
# (never with other trace domains), each followed by the known-byte-count calibration of tools/pmc_calib.py.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS="--no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras"
db() { find $1 -name '*.db' | head -1; }

timeout 900 python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench.err
python - <<EOF2
import json
d = json.load(open('$OUT/${TAG}_bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline frac', d['roofline']['frac'], 'hbm_frac', d['roofline'].get('hbm_frac'))
EOF2

# the round's changes on the same box, interleaved (spg_tune key 22: the pooled layer's backward as the separate weight- / data-gradient
# launches of round 5 instead of the two-pass fused pair; key 14: no fused backward pairs at all; key 21: the optimiser's fail-safe read of
# the recurrences' time-out word off; 17 / 18: the one-pass first layers off)
for i in 1 2; do
for V in "--tune 14:1" "--tune 22:1" "" "--tune 21:1" "--tune 17:1,18:1" "--sync-bn 1"; do
  timeout 300 python $ROOT/bench.py --steps 40 --warmup 10 $STEPS $V 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$V]', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'median', round(d['ms_per_step_median'],4), d['config']['batchnorm'])"
done
done > $OUT/${TAG}_ab_step_paths.txt
cat $OUT/${TAG}_ab_step_paths.txt

timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace -- python $ROOT/bench.py --steps 20 --warmup 5 $STEPS > /dev/null 2> $OUT/${TAG}_trace.err
python $ROOT/tools/prof_summary.py $(db /tmp/p_trace) 70 > $OUT/${TAG}_kernel_stats.txt
python $ROOT/tools/prof_timeline.py $(db /tmp/p_trace) 25 > $OUT/${TAG}_timeline.txt
head -5 $OUT/${TAG}_kernel_stats.txt
for S in 2 8; do
  timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace_s$S -- python $ROOT/bench.py --scenes $S --steps 10 --warmup 4 $STEPS > /dev/null 2>> $OUT/${TAG}_trace.err
  python $ROOT/tools/prof_summary.py $(db /tmp/p_trace_s$S) 40 > $OUT/${TAG}_kernel_stats_scenes$S.txt
done
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace_x3 -- python $ROOT/bench.py --precision bf16x3 --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 --steps 8 --warmup 3 $STEPS > /dev/null 2>> $OUT/${TAG}_trace.err
python $ROOT/tools/prof_summary.py $(db /tmp/p_trace_x3) 40 > $OUT/${TAG}_kernel_stats_sema3d_bf16x3.txt

for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format rocpd -d /tmp/p_$C -- python $ROOT/bench.py --steps 4 --warmup 2 $STEPS > /dev/null 2> $OUT/${TAG}_pmc_$C.err
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format rocpd -d /tmp/c_$C -- python $ROOT/tools/pmc_calib.py 1024 > /dev/null 2>> $OUT/${TAG}_pmc_$C.err
  python $ROOT/tools/pmc_summary.py $(db /tmp/p_$C) 24 > $OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z).txt
  python $ROOT/tools/pmc_summary.py $(db /tmp/c_$C) 0 > $OUT/${TAG}_pmc_calib_$(echo $C | tr A-Z a-z).txt
done
python $ROOT/tools/pmc_traffic.py $(db /tmp/p_FETCH_SIZE) $(db /tmp/p_WRITE_SIZE) $(db /tmp/c_FETCH_SIZE) $(db /tmp/c_WRITE_SIZE) 1024 > $OUT/${TAG}_gemm_traffic.json
head -c 800 $OUT/${TAG}_gemm_traffic.json

# MFMA utilisation evidence: SQ counters of the same command (own pass, no other trace domains)
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format rocpd -d /tmp/p_sq_f32 -- python $ROOT/bench.py --steps 4 --warmup 2 $STEPS > /dev/null 2> $OUT/${TAG}_pmc_sq_f32.err
python $ROOT/tools/pmc_summary.py $(db /tmp/p_sq_f32) 24 > $OUT/${TAG}_pmc_sq_counters_f32.txt
# host enqueue time of a step, module path and spg_train_step
timeout 300 python $ROOT/tools/host_time.py 2>&1 | grep -v "^ \|^$\|function calls\|Ordered\|List reduced\|ncalls" > $OUT/${TAG}_host_enqueue.txt
cat $OUT/${TAG}_host_enqueue.txt
