#!/bin/bash
# Everything the round's measurement deliverables need, in one GPU-box call (run from the repo root):
#   tools/collect_profiles.sh <tag>            e.g. r03  ->  gpurun_out/<tag>_{bench.json,kernel_stats.txt,pmc_*.txt,gemm_traffic.json}
# Passes: (1) bench.py (default flags: the driver's command), (2) rocprofv3 --kernel-trace of the train steps, summarised
# over the steady-state window (tools/prof_summary.py), (3)+(4) --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes
# (never with other trace domains), each followed by the known-byte-count calibration of tools/pmc_calib.py.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS="--no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras"
db() { find $1 -name '*.db' | head -1; }

timeout 600 python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 2000 $OUT/${TAG}_bench.json

# the opt-in precision modes: separate lines, same box, same session (they never replace the f32 headline)
for P in bf16x3 bf16; do
  timeout 300 python $ROOT/bench.py --precision $P --no-cpu-baseline --no-trainer-window --no-forward-only --no-live-pmc > $OUT/${TAG}_bench_$P.json 2>> $OUT/${TAG}_bench.err
done
timeout 300 python $ROOT/bench.py --precision bf16x3 --no-cpu-baseline --no-trainer-window --no-forward-only --no-live-pmc --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 > $OUT/${TAG}_bench_sema3d_bf16x3.json 2>> $OUT/${TAG}_bench.err
timeout 300 python $ROOT/bench.py --precision f32 --no-cpu-baseline --no-trainer-window --no-forward-only --no-live-pmc --n-sp 10000 --n-edges 50000 --n-feat 11 --model-config gru_10,f_8 > $OUT/${TAG}_bench_sema3d_f32.json 2>> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace_x3 -- python $ROOT/bench.py --precision bf16x3 --steps 20 --warmup 5 $STEPS > /dev/null 2> $OUT/${TAG}_trace_x3.err
python $ROOT/tools/prof_summary.py $(db /tmp/p_trace_x3) 70 > $OUT/${TAG}_kernel_stats_bf16x3.txt

timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_trace -- python $ROOT/bench.py --steps 20 --warmup 5 $STEPS > /dev/null 2> $OUT/${TAG}_trace.err
python $ROOT/tools/prof_summary.py $(db /tmp/p_trace) 70 > $OUT/${TAG}_kernel_stats.txt
head -5 $OUT/${TAG}_kernel_stats.txt

for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format rocpd -d /tmp/p_$C -- python $ROOT/bench.py --steps 4 --warmup 2 $STEPS > /dev/null 2> $OUT/${TAG}_pmc_$C.err
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format rocpd -d /tmp/c_$C -- python $ROOT/tools/pmc_calib.py 1024 > /dev/null 2>> $OUT/${TAG}_pmc_$C.err
  python $ROOT/tools/pmc_summary.py $(db /tmp/p_$C) 20 > $OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z).txt
  python $ROOT/tools/pmc_summary.py $(db /tmp/c_$C) 0 > $OUT/${TAG}_pmc_calib_$(echo $C | tr A-Z a-z).txt
done
python $ROOT/tools/pmc_traffic.py $(db /tmp/p_FETCH_SIZE) $(db /tmp/p_WRITE_SIZE) $(db /tmp/c_FETCH_SIZE) $(db /tmp/c_WRITE_SIZE) 1024 > $OUT/${TAG}_gemm_traffic.json
head -c 1500 $OUT/${TAG}_gemm_traffic.json

# MFMA utilisation evidence: SQ counters of the same command (own pass, no other trace domains), f32 and split-bf16
for P in f32 bf16x3; do
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format rocpd -d /tmp/p_sq_$P -- python $ROOT/bench.py --precision $P --steps 4 --warmup 2 $STEPS > /dev/null 2> $OUT/${TAG}_pmc_sq_$P.err
  python $ROOT/tools/pmc_summary.py $(db /tmp/p_sq_$P) 20 > $OUT/${TAG}_pmc_sq_counters_$P.txt
done
