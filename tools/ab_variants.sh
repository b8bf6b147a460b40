#!/bin/bash
# same-box A/B of library variants built by tools/build_variant.sh:  tools/ab_variants.sh <tag> <reps> <name1> <name2> ...   ("base" etc.)
# per variant: tests/test_gpu_bwdpair.py (correctness gate), <reps> interleaved bench runs, one kernel trace (per-kernel averages)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=$2; shift 2
OUT=$ROOT/gpurun_out; mkdir -p $OUT
V=superpoint_graph_amd/csrc/variants
LIBS=""
for n in "$@"; do LIBS="$LIBS $V/libspg_$n.so"; done
{
for n in "$@"; do
  echo "== $n: pytest ${AB_TESTS:-tests/test_gpu_bwdpair.py}"
  SPG_HIP_LIB=$ROOT/$V/libspg_$n.so timeout 900 python -m pytest ${AB_TESTS:-tests/test_gpu_bwdpair.py} -x -q -m gpu 2>&1 | tail -3
done
echo "== interleaved bench ($N reps) $AB_ARGS"
bash $ROOT/tools/ab_libs.sh $N $LIBS
for n in "$@"; do
  echo "== $n: kernel trace"
  SPG_HIP_LIB=$ROOT/$V/libspg_$n.so bash $ROOT/tools/quick_stats.sh ${TAG}_$n $AB_ARGS > /dev/null 2>&1
  head -${AB_TOP:-28} $OUT/${TAG}_${n}_kernel_stats.txt | cut -c1-150
done
} 2>&1 | tee $OUT/${TAG}_ab.txt
