// Internal interface of the narrow-layer kernels (spg_narrow.hip): the first two convolutions of a PointNet segment in one pass.
#pragma once
#include "spg_gemm.h"

#define SPG_GRAM_MAXF 32      // point features (the cloud path of the general kernels has the same bound)
// statistics slots of the Gram matrix of [x; 1]: [SPG_FOLD_SLOTS][2 limbs: hi, lo][npairs], npairs = (nfeat + 1)(nfeat + 2) / 2
// (upper triangle, row-major), then a flag word -- same fixed-point format as the layers' slots (spg_fold.h, SH = -8)
inline size_t spg_gram_slot_words(int nfeat) { const int c = nfeat + 1; return (size_t)SPG_FOLD_SLOTS * 2 * (c * (c + 1) / 2) + 8; }

struct SpgGramParams {
  const float* clouds;        // [B, Ctot, P] channel-major
  const float* stnT;          // [B, 4] raw STN projection (identity added by the kernel), or null
  int B, P, Ctot, nfeat;
  unsigned long long* gram;   // zero before the launch
};
int spg_launch_cloud_gram(const SpgGramParams& p, hipStream_t stream);

struct SpgNarrowPairParams {
  const float* clouds; const float* stnT;
  int P, Ctot, nfeat, nblk;   // nblk = B * P / 32 blocks of 32 points
  double count;               // B * P: rows behind the first layer's statistics
  // first layer: y1 = W1 x + b1, train-mode BatchNorm from the Gram matrix
  const float *W1, *b1;       // [64, nfeat], [64] or null
  float* y1;                  // out [B * P, 64]
  const unsigned long long* gram;
  const float *gamma1, *beta1;
  float *rm1, *rv1;           // running statistics (may be null)
  float *mean1, *rstd1, *s1, *t1;   // out [64] each (workgroup 0)
  int update_times;
  float momentum, eps;
  // second layer: y2 = W2 relu(s1 y1 + t1) + b2, its statistics into slots2
  const float *W2, *b2;       // [64, 64], [64] or null
  float* y2;                  // out [B * P, 64]
  unsigned long long* slots2;
};
bool spg_narrow_pair_supported(int nfeat, int c1, int c2, int P, long M);
int spg_launch_narrow_pair_fwd(const SpgNarrowPairParams& p, hipStream_t stream);
