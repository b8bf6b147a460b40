// Internal interface of the narrow-layer kernels (spg_narrow.hip): the first two convolutions of a PointNet segment in one pass.
#pragma once
#include "spg_gemm.h"

#define SPG_GRAM_MAXF 32      // point features (the cloud path of the general kernels has the same bound)
// statistics slots of the Gram matrix of [x; 1]: [SPG_FOLD_SLOTS][2 limbs: hi, lo][npairs], npairs = (nfeat + 1)(nfeat + 2) / 2
// (upper triangle, row-major), then a flag word and the row count -- same fixed-point format as the layers' slots (spg_fold.h, SH = -8)
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

// ---- backward of a segment's FIRST convolution without a pass over its raw output (round 5) ----------------------------------
// dW1 = sum_rows dz1^T x and (main segment behind an STN) dT = sum_points x_raw (dz1 W1[:, 0:2]) used to be two row-GEMM launches
// over (g, y1) -- 66 MB each -- plus spg_stn_dT.  With dz1 = s (g - c1) - d (y1 - mu) (BatchNorm backward; g = the masked data
// gradient the 64 -> 64 layer's fused backward wrote, s / mu the forward constants, c1 = mean g, d = s c2 rstd) and y1 = W1 x + b1
// LINEAR in the cloud, everything that involves y1 collapses onto the Gram matrix G of the input (spg_cloud_gram*, already in the
// statistics slots).  With the input centred, x'' = x - mean x (so that sum x'' = 0: no large cancelling terms):
//   dW1[c, k]  = s_c sum_rows g[row, c] x''[row, k]  -  d_c sum_j W1[c, j] Gc[j, k]             (Gc = G - M xbar xbar^T)
//   dxy[row, n] = sum_c s_c W1[c, n] g[row, c]  -  sum_c s_c c1_c W1[c, n]  -  x''[row] . v_n,   v_n[k] = sum_c W1[c, k] d_c W1[c, n]
// ONE pass over g (33 MB) and the cloud (7 MB): every wavefront takes the blocks of 32 points of a superpoint, accumulates
// x''^T g on the matrix pipe over all its blocks, forms dxy per point with a transposing butterfly and the 2 x 2 gradient dT of
// its superpoint in registers.  y1 is not read; spg_stn_dT and the dxy buffer are gone.
struct SpgFirstConvBwdParams {
  const float* clouds; const float* stnT;   // the segment's input (as the forward read it)
  int B, P, Ctot, nfeat;
  const float* g;                            // [B * P, 64]: masked gradient wrt the first layer's BatchNorm output (ld 64)
  const float* W1;                           // [64, nfeat]
  const unsigned long long* gram;            // the forward's Gram slots of this segment (still intact)
  SpgBnFoldBwd fold;                         // the first layer's BatchNorm-backward sums -> constants, dgamma / dbeta
  float* partial;                            // out [spg_first_conv_bwd_partials(B)][64 * nfeat]: partials of dW1 for the batched reduction
  float* dT;                                 // out [B, 4] or null
};
bool spg_first_conv_bwd_supported(int nfeat, int c1, int P, long M);
int spg_first_conv_bwd_partials(int B);
int spg_launch_first_conv_bwd(const SpgFirstConvBwdParams& p, hipStream_t stream);
