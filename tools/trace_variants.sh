#!/bin/bash
# kernel-trace averages of selected kernels for several library variants (timing attribution; no correctness gate):
#   AB_GREP="bwdpair" tools/trace_variants.sh <tag> <name1> <name2> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out; mkdir -p $OUT
V=superpoint_graph_amd/csrc/variants
for n in "$@"; do
  echo "== $n"
  SPG_HIP_LIB=$ROOT/$V/libspg_$n.so bash $ROOT/tools/quick_stats.sh ${TAG}_$n $AB_ARGS > /dev/null 2>&1
  head -1 $OUT/${TAG}_${n}_kernel_stats.txt | cut -c1-160
  grep -E "${AB_GREP:-bwdpair}" $OUT/${TAG}_${n}_kernel_stats.txt | cut -c1-150
done 2>&1 | tee $OUT/${TAG}_trace_variants.txt
