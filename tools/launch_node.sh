#!/bin/bash
# Data-parallel bench on one node: one process per GPU, scenes sharded one per rank, RCCL over xGMI.
#   tools/launch_node.sh [N_GPUS=8] [SYNC_BN=0] [extra bench.py flags]
# Runs N = 1, 2, 4, ... up to N_GPUS back to back when N_GPUS is "sweep".  The collectives (one flat fp32 gradient
# all-reduce per step; with SYNC_BN=1 also 26 small fp64 BatchNorm all-reduces) are issued by libspg_hip's own RCCL
# communicator when SYNC_BN=1 or `--native-rccl 1` is passed, else by torch.distributed (bench.py --native-rccl).
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
N="${1:-8}"; SYNC="${2:-0}"; shift $(( $# > 2 ? 2 : $# )) || true
run() {
  local n="$1"; shift
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --sync-bn "$SYNC" "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus "$n" --sync-bn "$SYNC" --no-forward-only --no-cpu-baseline "$@"
  fi
}
if [ "$N" = "sweep" ]; then
  AVAIL=$(python -c 'import torch; print(torch.cuda.device_count())')
  for n in 1 2 4 8; do [ "$n" -le "$AVAIL" ] && run "$n" "$@"; done
else
  run "$N" "$@"
fi
