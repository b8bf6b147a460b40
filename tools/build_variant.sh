#!/bin/bash
# build a VARIANT of libspg_hip.so for same-box A/Bs (tools/ab_libs.sh): tools/build_variant.sh <name> "<extra hipcc flags>" [sources to recompile ...]
# only the listed sources (default: spg_gemm.hip) are recompiled with the extra flags; the other objects come from the in-tree build
set -e
cd "$(dirname "$0")/../superpoint_graph_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
SRCS=${@:-spg_gemm.hip}
mkdir -p variants
OBJS=""
for S in spg_gemm spg_ecc spg_api spg_pointnet spg_eccnet spg_loader spg_convstack spg_batch spg_rccl spg_loss spg_spgraph spg_step spg_narrow; do
  if echo " $SRCS " | grep -q " $S.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c $S.hip -o variants/${S}_$NAME.o &
    OBJS="$OBJS variants/${S}_$NAME.o"
  else
    OBJS="$OBJS $S.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o variants/libspg_$NAME.so
ls -la variants/libspg_$NAME.so
