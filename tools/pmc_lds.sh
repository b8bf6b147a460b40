ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
STEPS="--no-cpu-baseline --no-forward-only --no-trainer-window --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format rocpd -d /tmp/p_lds -- python $ROOT/bench.py --steps 4 --warmup 2 $STEPS > /dev/null 2> $ROOT/gpurun_out/r04_pmc_lds.err
python $ROOT/tools/pmc_summary.py $(find /tmp/p_lds -name '*.db' | head -1) 24 > $ROOT/gpurun_out/r04_pmc_lds_counters.txt
tail -3 $ROOT/gpurun_out/r04_pmc_lds.err; grep -E "bwdpair|128, 128, 2, 2, (false, 1|true, 4)|wgrad_kernel<128, 128, 2, 2, 4" $ROOT/gpurun_out/r04_pmc_lds_counters.txt | cut -c1-400
