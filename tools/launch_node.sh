#!/bin/bash
# Data-parallel bench on one node: one process per GPU, scenes sharded one per rank, ONE flat fp32 gradient all-reduce per step
# over RCCL / xGMI (torch.distributed's communicator; with SYNC_BN=1 also the 13 BatchNorm layers' per-channel fp64 sums, forward
# and backward).  `--native-rccl 1` (extra flag) lets libspg_hip issue the collectives through its OWN RCCL communicator instead --
# opt-in: that communicator has only ever run at world size 1 (tests/test_gpu_dist.py runs it at world size 2 whenever a box has
# two GPUs).
#   tools/launch_node.sh [N_GPUS=8 | sweep] [SYNC_BN=0] [extra bench.py flags]
# "sweep": N = 1, 2, 4, 8 (up to the GPUs present) back to back; every run prints exactly ONE JSON line in bench.py's schema on
# stdout (value = superpoints/s of the whole job, weak scaling: one scene per rank), i.e. the lines of a SCALE record.
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
N="${1:-8}"; SYNC="${2:-0}"; shift $(( $# > 2 ? 2 : $# )) || true
SIDE="--no-forward-only --no-cpu-baseline --no-trainer-window --no-extras --no-live-pmc"
run() {
  local n="$1"; shift
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --sync-bn "$SYNC" $SIDE "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus "$n" --sync-bn "$SYNC" $SIDE "$@"
  fi
}
if [ "$N" = "sweep" ]; then
  AVAIL=$(python -c 'import torch; print(torch.cuda.device_count())')
  for n in 1 2 4 8; do [ "$n" -le "$AVAIL" ] && run "$n" "$@"; done
else
  run "$N" "$@"
fi
