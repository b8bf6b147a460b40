// The first two convolutions of a PointNet segment (raw cloud -> 64 -> 64 channels; learning/pointnet.py:84-96, STNkD :31-37) in
// train mode as ONE pass over the points (round 5).
//
// As two launches of the general row-GEMM they ran at 2.0 / 2.7 TB/s -- neither the HBM nor the MFMA roof: each workgroup sees one
// or two 128-row tiles, so the launch is a sequence of exposed latencies (first loads, LDS staging behind workgroup barriers,
// the epilogue), and the 33 MB of the first layer's raw output make a round trip through HBM in between.  What forces two
// launches is train-mode BatchNorm: the second layer needs the first layer's batch statistics.  But the first layer is LINEAR
// in the cloud, y1 = W1 x + b1, so its per-channel statistics follow from the (nfeat+1) x (nfeat+1) Gram matrix of the
// augmented input [x; 1] over all points:   sum y1_c = w~_c . G[:, last],   sum y1_c^2 = w~_c^T G w~_c,   w~_c = [W1[c, :], b1[c]].
//   * spg_cloud_gram_kernel: one workgroup per superpoint, G in float64 (products of two floats are exact in float64), added to
//     the layer's statistics slots as exact fixed-point integers (spg_fold.h) -- 7 MB read, ~4 us;
//   * spg_narrow_pair_fwd_kernel: every WAVEFRONT takes blocks of 32 points through BOTH layers on its own: cloud channels straight
//     from global memory into MFMA A operands (channel-major clouds are coalesced along the points = along the lanes), y1 =
//     conv1 (written once, for the backward), BatchNorm + ReLU with the constants every workgroup derives from G in its
//     prologue, a wave-private LDS tile to turn the accumulator layout into the A operand of conv2, y2 = conv2 (W2 resident in
//     LDS), statistics of y2 merged over the wave's blocks into ONE fixed-point contribution.  No workgroup barrier after the
//     prologue; 16 waves per CU.  HBM: x in, y1 and y2 out -- y1 is never read back.
// The MFMA order over the reduction index is that of the general kernels (spg_common.h: spg_mfma_chunk), and so are the
// expressions of the STN transform, the affine + ReLU and the statistics fold; what differs from the two-launch path is the
// first layer's statistics (exact instead of accumulated from rounded fp32 outputs: ~1e-7 relative) and the partition of
// the second layer's statistics partials -- hence parity at tolerance, not bit-identity, with spg_tune key 17 = 1 (two launches).
#include "../../include/spg_hip.h"
#include "spg_fold.h"
#include "spg_narrow.h"

#pragma clang fp contract(off)

namespace {

#define SPG_NP_C 64                 // output channels of both layers
// stores of the two raw outputs (y1 is read again by the backward pass only, y2 by the next layer and by the backward pass): NON-TEMPORAL,
// as the row-GEMMs' outputs (spg_gemm.hip: spg_store_tile_vec_impl) -- same box, interleaved: y1 -0.2 %, y1 and y2 -0.55 % of the step
#define SPG_NP_ST1(ptr, v) __builtin_nontemporal_store((v), (ptr))
#define SPG_NP_ST2(ptr, v) __builtin_nontemporal_store((v), (ptr))
#define SPG_NP_W1LD 33              // floats per row of the W1 copy in LDS (<= 32 input channels + pad)
#define SPG_NP_A_SLOTS (16 * 33)    // float4 slots of one wave's activation tile: 16 planes (64 channels / 4) x (32 rows + 1)
#define SPG_NP_W2_SLOTS (16 * 65)   // float4 slots of W2 in out-major layout: 16 planes x (64 columns + 1)

// ---- Gram matrix of the augmented input over all points of the batch ----
__global__ __launch_bounds__(256) void spg_cloud_gram_kernel(const SpgGramParams p) {
  __shared__ float xs[128 * (SPG_GRAM_MAXF + 1)];
  const int g = blockIdx.x, P = p.P, nf = p.nfeat, Cg = nf + 1, tid = threadIdx.x;
  const float* base = p.clouds + (long)g * p.Ctot * P;
  for (int i = tid; i < nf * P; i += 256) {
    const int c = i / P, pt = i - c * P;
    xs[pt * Cg + c] = base[i];
  }
  for (int pt = tid; pt < P; pt += 256) xs[pt * Cg + nf] = 1.f;
  __syncthreads();
  if (p.stnT != nullptr) {      // learning/pointnet.py:123  [x y] @ (proj.view(2,2) + I): the expressions of spg_fetch (spg_common.h)
    const float* T = p.stnT + (long)g * 4;
    const float t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3];
    for (int pt = tid; pt < P; pt += 256) {
      const float x = xs[pt * Cg], y = xs[pt * Cg + 1];
      xs[pt * Cg] = fmaf(x, t0 + 1.f, y * t2);
      xs[pt * Cg + 1] = fmaf(x, t1, y * (t3 + 1.f));
    }
    __syncthreads();
  }
  const int npairs = Cg * (Cg + 1) / 2;
  if (blockIdx.x == 0 && tid == 0)      // the rows behind the sums travel with them (spg_fold.h: spg_slots_count_add)
    __hip_atomic_fetch_add(p.gram + (size_t)SPG_FOLD_SLOTS * 2 * npairs + 1, (unsigned long long)((long)p.B * P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long* slot = p.gram + (size_t)(blockIdx.x & (SPG_FOLD_SLOTS - 1)) * 2 * npairs;
  for (int pr = tid; pr < npairs; pr += 256) {
    int i = 0, rem = pr;
    while (rem >= Cg - i) { rem -= Cg - i; ++i; }
    const int j = i + rem;
    double acc = 0.0;
    for (int pt = 0; pt < P; ++pt) acc = fma((double)xs[pt * Cg + i], (double)xs[pt * Cg + j], acc);
    if (!(fabs(acc) <= 0x1p52)) atomicOr(p.gram + (size_t)SPG_FOLD_SLOTS * 2 * npairs, 1ull);      // NaN / inf / out of range
    long long hi, lo;
    spg_fx_split<-8>(acc, hi, lo);
    __hip_atomic_fetch_add(slot + pr, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(slot + npairs + pr, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ double spg_shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}

// The same Gram matrix for nfeat + 1 <= 16 (every documented configuration: 14 / 11 / 9 features) on the float64 matrix pipe:
// ONE WAVEFRONT per superpoint, no LDS, no barrier.  G = X^T X with X = [points x 16 augmented channels] is the product the
// instruction v_mfma_f64_16x16x4_f64 computes when BOTH operands are the lane's own value: lane l supplies A[i = l % 16][k = l / 16]
// and B[k = l / 16][j = l % 16], i.e. channel l % 16 of point k.  The points of a superpoint are consumed 4 per instruction in any
// order (a sum over all of them): lane (i, kq) loads its channel's points 16 u + 4 kq + {0..3} as one float4 (u = 0 .. P/16 - 1) and
// instruction (u, e) takes element e.  The result D[i][j] (lane l, register v: i = l / 16 + 4 v, j = l % 16 -- pinned by
// tools/probe/mfma_f64_probe.hip) goes to the
// fixed-point slots for i <= j.  The 2 x 2 STN transform needs x and y of a point in one lane: neighbouring lanes exchange them.
// the value of the neighbouring lane (lane ^ 1): a quad permutation on the vector ALU instead of a trip through the LDS crossbar
__device__ __forceinline__ float spg_lane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define SPG_GRAM16_SPW 2      // superpoints per wavefront (their points extend ONE reduction)
__global__ __launch_bounds__(256) void spg_cloud_gram16_kernel(const SpgGramParams p) {
  __shared__ double red[3 * 256];      // the accumulators of waves 1 .. 3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4, nf = p.nfeat, P = p.P, Cg = nf + 1;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  const int g0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * 4 + wave) * SPG_GRAM16_SPW);
  f32x4 x[SPG_GRAM16_SPW][8], T[SPG_GRAM16_SPW];
#pragma unroll
  for (int q = 0; q < SPG_GRAM16_SPW; ++q) {      // every load of the wave in flight first
    const int g = g0 + q < p.B ? g0 + q : p.B - 1;
    const float* row = p.clouds + ((long)g * p.Ctot + (i < nf ? i : 0)) * P + 4 * kq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x[q][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (16 * u < P) x[q][u] = *reinterpret_cast<const f32x4*>(row + 16 * u);      // (P is a multiple of 16: whole float4 groups)
    }
    T[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.stnT != nullptr) T[q] = *reinterpret_cast<const f32x4*>(p.stnT + (long)g * 4);
  }
#pragma unroll
  for (int q = 0; q < SPG_GRAM16_SPW; ++q) {
    if (g0 + q >= p.B) break;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (16 * u >= P) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = x[q][u][e];
        if (p.stnT != nullptr) {      // lanes i = 0 (x) and i = 1 (y) are neighbours: learning/pointnet.py:123, the expressions of spg_fetch
          const float other = spg_lane_xor1(v);
          if (i == 0) v = fmaf(v, T[q][0] + 1.f, other * T[q][2]);
          else if (i == 1) v = fmaf(other, T[q][1], v * (T[q][3] + 1.f));
        }
        v = i < nf ? v : (i == nf ? 1.f : 0.f);
        const double d = (double)v;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(d, d, acc, 0, 0, 0);
      }
    }
  }
  // ONE contribution per workgroup and entry: device-scope atomics on one address are served one after the other
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) red[(wave - 1) * 256 + v * 64 + lane] = acc[v];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] += red[w * 256 + v * 64 + lane];      // fixed order: deterministic
  const int npairs = Cg * (Cg + 1) / 2;
  if (blockIdx.x == 0 && lane == 0)     // the rows behind the sums travel with them (spg_fold.h: spg_slots_count_add)
    __hip_atomic_fetch_add(p.gram + (size_t)SPG_FOLD_SLOTS * 2 * npairs + 1, (unsigned long long)((long)p.B * P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long* slot = p.gram + (size_t)(blockIdx.x & (SPG_FOLD_SLOTS - 1)) * 2 * npairs;
  const int j = lane & 15;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int ii = (lane >> 4) + 4 * v;
    if (ii <= j && j < Cg) {
      const int pr = ii * Cg - ii * (ii - 1) / 2 + (j - ii);
      const double a = acc[v];
      if (!(fabs(a) <= 0x1p52)) atomicOr(p.gram + (size_t)SPG_FOLD_SLOTS * 2 * npairs, 1ull);
      long long hi, lo;
      spg_fx_split<-8>(a, hi, lo);
      __hip_atomic_fetch_add(slot + pr, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(slot + npairs + pr, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// KG: reduction groups of 8 input channels of the first layer (2: nfeat <= 16, 4: nfeat <= 32); NW: wavefronts per workgroup
// The cloud channels of one block for lane (r, h): channels 8 gq + 4 h + s of point p0 + r = the A operand of MFMA (gq, s)
template <int KG>
__device__ __forceinline__ void spg_np_load_x(const SpgNarrowPairParams& p, int blk, int bps, int r, int h, float (&v)[4 * KG], f32x4& T) {
  const int g = blk / bps, p0 = (blk - g * bps) * 32;
  // (32-bit element offset from the wave-uniform tensor base; the launcher checks that the tensor is below 2^32 bytes)
  const unsigned xo = (unsigned)(g * p.Ctot * p.P + p0 + r);
#pragma unroll
  for (int gq = 0; gq < KG; ++gq)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 8 * gq + 4 * h + s;
      const float t = p.clouds[xo + (unsigned)((c < p.nfeat ? c : 0) * p.P)];      // (unconditional load + select: no branches)
      v[gq * 4 + s] = c < p.nfeat ? t : 0.f;
    }
  T = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.stnT != nullptr) T = *reinterpret_cast<const f32x4*>(p.stnT + (long)g * 4);
}

template <int KG, int NW>
__global__ __launch_bounds__(64 * NW) void spg_narrow_pair_fwd_kernel(const SpgNarrowPairParams p) {
  extern __shared__ f32x4 smem[];
  f32x4* W2s = smem;
  float* W1s = reinterpret_cast<float*>(smem + SPG_NP_W2_SLOTS + NW * SPG_NP_A_SLOTS);      // [64][SPG_NP_W1LD]: column nfeat holds the bias
  float* cst = W1s + SPG_NP_C * SPG_NP_W1LD;                                                // s1 [64], t1 [64]
  double* Gs = reinterpret_cast<double*>(smem + SPG_NP_W2_SLOTS);                           // prologue only: [Cg][Cg], over the wave tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nf = p.nfeat, Cg = nf + 1, npairs = Cg * (Cg + 1) / 2;
  constexpr int NT = 64 * NW;
  const int bps = p.P / 32;                       // blocks per superpoint
  const int total = (int)gridDim.x * NW;
  int blk = (int)blockIdx.x * NW + wave;

  // the first block's input and every small per-channel operand are on their way while the prologue runs (dependent global-load
  // round trips in the prologue were ~1.5 us each)
  float v[4 * KG];
  f32x4 T;
  spg_np_load_x<KG>(p, blk < p.nblk ? blk : 0, bps, r, h, v, T);
  float b2c[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) b2c[j] = p.b2 != nullptr ? p.b2[r + 32 * j] : 0.f;
  const int c = tid / NW, part = tid - c * NW;      // channel c: NW consecutive lanes, rows i = part, part + NW, ... of the quadratic form
  const float ga1 = p.gamma1 ? p.gamma1[c] : 1.f, be1 = p.beta1 ? p.beta1[c] : 0.f;
  float rm1 = 0.f, rv1 = 0.f;
  if (p.rm1 != nullptr && part == 0 && blockIdx.x == 0) { rm1 = p.rm1[c]; rv1 = p.rv1[c]; }
  // (the Gram slots' flag and row-count words as well: the count used to be loaded behind the barrier below -- one more round trip)
  const unsigned long long gram_bad = p.gram[(size_t)SPG_FOLD_SLOTS * 2 * npairs];
  const unsigned long long nrow = p.gram[(size_t)SPG_FOLD_SLOTS * 2 * npairs + 1];      // rows of all ranks (slot-synchronised BatchNorm), else this rank's

  // ---- prologue: weights into LDS; the first layer's BatchNorm constants from the Gram matrix ----
  for (int i = tid; i < SPG_NP_C * 16; i += NT) {
    const int col = i >> 4, q = i & 15;
    W2s[q * 65 + col] = *reinterpret_cast<const f32x4*>(p.W2 + (long)col * SPG_NP_C + 4 * q);
  }
  {   // (unconditional clamped loads, all in flight, then the selects: a conditional load per element kept the trips of this loop in order)
    constexpr int NE1 = SPG_NP_C * SPG_NP_W1LD, NI1 = (NE1 + NT - 1) / NT;
    float wv1[NI1], bv1[NI1];
#pragma unroll
    for (int u = 0; u < NI1; ++u) {
      const int i = min(tid + u * NT, NE1 - 1), col = i / SPG_NP_W1LD, k = i - col * SPG_NP_W1LD;
      wv1[u] = p.W1[(long)col * nf + (k < nf ? k : 0)];
      bv1[u] = p.b1 != nullptr ? p.b1[col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NI1; ++u) {
      const int i = tid + u * NT, col = i / SPG_NP_W1LD, k = i - col * SPG_NP_W1LD;
      if (i < NE1) W1s[i] = k < nf ? wv1[u] : (k == nf ? bv1[u] : 0.f);
    }
  }
  {
    const bool bad = gram_bad != 0ull;
    for (int pr = tid; pr < npairs; pr += NT) {
      int i = 0, rem = pr;
      while (rem >= Cg - i) { rem -= Cg - i; ++i; }
      const int j = i + rem;
      double g = spg_fx_sum<-8>(p.gram + pr, (size_t)npairs, (size_t)2 * npairs);
      if (bad) g = __builtin_nan("");
      Gs[i * Cg + j] = g;
      Gs[j * Cg + i] = g;
    }
  }
  __syncthreads();
  double mean = 0.0, m2 = 0.0;
  float s1v = 0.f, t1v = 0.f;
  {
    double lin = 0.0, quad = 0.0;
    for (int i = part; i < Cg; i += NW) {
      const double wi = (double)W1s[c * SPG_NP_W1LD + i];      // (w~ = [W1[c, :], b1[c]]: the LDS copy keeps the bias in column nfeat)
      double row = 0.0;
      for (int j = 0; j < Cg; ++j) row = fma((double)W1s[c * SPG_NP_W1LD + j], Gs[i * Cg + j], row);
      quad = fma(wi, row, quad);
      lin = fma(wi, Gs[i * Cg + nf], lin);
    }
#pragma unroll
    for (int m = NW / 2; m >= 1; m >>= 1) { lin += spg_shfl_xor_d(lin, m); quad += spg_shfl_xor_d(quad, m); }
    // (the arithmetic of spg_bn_fold_fwd, spg_gemm.hip)
    const double M = nrow != 0ull ? (double)nrow : p.count;
    mean = lin / M;
    m2 = quad - M * mean * mean;
    if (m2 < 0.0) m2 = 0.0;
    const double rstd = 1.0 / sqrt(m2 / M + (double)p.eps);
    const double ga = (double)ga1, be = (double)be1;
    s1v = (float)(ga * rstd); t1v = (float)(be - mean * ga * rstd);
    if (part == 0 && blockIdx.x == 0) {
      p.mean1[c] = (float)mean; p.rstd1[c] = (float)rstd; p.s1[c] = s1v; p.t1[c] = t1v;
      if (p.rm1 != nullptr && p.update_times > 0) {
        const double var = m2 / M, uvar = M > 1.0 ? m2 / (M - 1.0) : var;
        float rm = rm1, rv = rv1;
        for (int u = 0; u < p.update_times; ++u) {
          rm = (1.f - p.momentum) * rm + p.momentum * (float)mean;
          rv = (1.f - p.momentum) * rv + p.momentum * (float)uvar;
        }
        p.rm1[c] = rm; p.rv1[c] = rv;
      }
    }
  }
  __syncthreads();      // (Gs lies over the wave tiles: everybody has read it)
  if (part == 0) { cst[c] = s1v; cst[SPG_NP_C + c] = t1v; }
  __syncthreads();      // the last workgroup barrier: from here on every wavefront is on its own

  f32x4* At = smem + SPG_NP_W2_SLOTS + wave * SPG_NP_A_SLOTS;
  float* Atf = reinterpret_cast<float*>(At);
  float s1c[2], t1c[2], b1c[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = r + 32 * j;
    s1c[j] = cst[col]; t1c[j] = cst[SPG_NP_C + col];
    b1c[j] = W1s[col * SPG_NP_W1LD + nf];      // (the LDS copy keeps the bias in column nfeat; 0 without a bias)
  }
  float sn = 0.f, sa[2] = {0.f, 0.f}, sb[2] = {0.f, 0.f};      // (rows, mean, M2) of this wave's blocks per column (Chan)
  const int lo = ((r >> 2) * 33 + 4 * h) * 4 + (r & 3);        // this lane's element of row 4h, column r in the LDS tile
  // Software pipeline over the wave's blocks.  A wave issues in order, so global stores placed BEHIND the MFMAs of a phase would be
  // issued with the matrix pipe idle, and (measured, profiles/r05_narrow_attribution.txt) the phases of a block then simply add up:
  // 15.6 us fixed + 7 us of conv2 + 6 us of stores.  Instead every store travels in the shadow of an MFMA of conv2: the 32 stores of
  // this block's y1 and the 32 of the PREVIOUS block's y2 (kept in a second register set) -- one store behind each of the 64 MFMAs.
  f32x16 yprev[2];                 // y2 of the previous block, not stored yet
  long yo_prev = -1;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) yprev[j][q] = 0.f;

  for (; blk < p.nblk; blk += total) {
    if (p.stnT != nullptr && h == 0) {      // learning/pointnet.py:123  [x y] @ (proj.view(2,2) + I): the expressions of spg_fetch
      const float x = v[0], y = v[1];
      v[0] = fmaf(x, T[0] + 1.f, y * T[2]);
      v[1] = fmaf(x, T[1], y * (T[3] + 1.f));
    }
    // ---- conv1 ----
    f32x16 acc[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < KG; ++gq)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = 8 * gq + 4 * h + s;
          const float w = W1s[(r + 32 * j) * SPG_NP_W1LD + k];
          // (the first product starts from the constant zero operand: no register clearing)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[gq * 4 + s], k < nf ? w : 0.f, (gq == 0 && s == 0) ? zero16 : acc[j], 0, 0, 0);
        }
    // the next block's input: in flight behind everything that follows
    {
      const int nb = blk + total;
      spg_np_load_x<KG>(p, nb < p.nblk ? nb : blk, bps, r, h, v, T);
    }
    // ---- y1 = conv1 + bias (kept in registers for the stores below); BatchNorm + ReLU into the wave's LDS tile in A-operand
    //      layout: accumulator register q of lane (r, h) is row (q & 3) + 8 (q >> 2) + 4 h, column r + 32 j (spg_acc_row) ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rl = (q & 3) + 8 * (q >> 2);
        acc[j][q] += b1c[j];
        Atf[lo + (8 * j * 33 + rl) * 4] = fmaxf(fmaf(acc[j][q], s1c[j], t1c[j]), 0.f);
      }
    __builtin_amdgcn_wave_barrier();      // (LDS executes a wave's instructions in order; this only pins the compiler's order)
    // (two row bases per tensor, 16 rows apart: every store is base + a compile-time offset that fits the instruction's immediate)
    const long yo = (long)blk * (32 * SPG_NP_C) + (4 * h * SPG_NP_C + r);      // rows 4h.., first half of the registers
    float* y1a = p.y1 + yo; float* y1b = y1a + 16 * SPG_NP_C;
    float* y2a = p.y2 + (yo_prev >= 0 ? yo_prev : yo); float* y2b = y2a + 16 * SPG_NP_C;
    const bool have_prev = yo_prev >= 0;      // wave-uniform
    // ---- conv2: accumulators start from the bias, reduction order of spg_mfma_chunk over the 16 planes ----
    f32x16 acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[j][q] = b2c[j];
    {
      // register double-buffering of the LDS fragments: the reads of group g + 1 are issued in front of the MFMAs of group g
      f32x4 a4[2], b4[2][2];
      a4[0] = At[h * 33 + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) b4[0][j] = W2s[h * 65 + r + 32 * j];
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) {
        const int cur = gq & 1, nxt = cur ^ 1;
        if (gq + 1 < 8) {
          a4[nxt] = At[(2 * (gq + 1) + h) * 33 + r];
#pragma unroll
          for (int j = 0; j < 2; ++j) b4[nxt][j] = W2s[(2 * (gq + 1) + h) * 65 + r + 32 * j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur][s], b4[cur][j][s], acc2[j], 0, 0, 0);
            // one store in the shadow of this MFMA: slots 0 .. 31 this block's y1, 32 .. 63 the previous block's y2
            const int slot = (gq * 4 + s) * 2 + j, e = slot & 31, ej = e >> 4, eq = e & 15, rl = (eq & 3) + 8 * (eq >> 2);
            if (slot < 32) {
              SPG_NP_ST1(eq < 8 ? y1a + rl * SPG_NP_C + 32 * ej : y1b + (rl - 16) * SPG_NP_C + 32 * ej, acc[ej][eq]);
            } else if (have_prev) {
              SPG_NP_ST2(eq < 8 ? y2a + rl * SPG_NP_C + 32 * ej : y2b + (rl - 16) * SPG_NP_C + 32 * ej, yprev[ej][eq]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- statistics of y2 (per column over the block's 32 rows, merged into the wave's running triple); y2 itself leaves with the
    //      next block's MFMAs ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc2[j][q];
      s += __shfl_xor(s, 32, 64);
      const float bmean = s * (1.f / 32.f);
      float bm2 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float d = acc2[j][q] - bmean;
        bm2 = fmaf(d, d, bm2);
      }
      bm2 += __shfl_xor(bm2, 32, 64);
      const float nn = sn + 32.f, f = 32.f / nn, delta = bmean - sa[j];
      sa[j] += delta * f;
      sb[j] += bm2 + delta * delta * (sn * f);
      yprev[j] = acc2[j];
    }
    sn += 32.f;
    yo_prev = yo;
  }
  if (yo_prev >= 0) {      // the last block's y2
    float* y2a = p.y2 + yo_prev; float* y2b = y2a + 16 * SPG_NP_C;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rl = (q & 3) + 8 * (q >> 2);
        SPG_NP_ST2(q < 8 ? y2a + rl * SPG_NP_C + 32 * j : y2b + (rl - 16) * SPG_NP_C + 32 * j, yprev[j][q]);
      }
  }
  // ONE statistics contribution per workgroup and column: the waves' triples meet in LDS (their tiles are free now) and wave 0
  // merges them in wave order (Chan) -- device-scope atomics on one address are served one after the other (~100 ns each): 2048
  // waves on 8 slots would queue 256 deep per address, 256 workgroups queue 32 deep
  __syncthreads();
  float* xch = reinterpret_cast<float*>(smem + SPG_NP_W2_SLOTS);      // [NW][64][2] + [NW] rows
  if (h == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xch[(wave * SPG_NP_C + r + 32 * j) * 2 + 0] = sa[j];
      xch[(wave * SPG_NP_C + r + 32 * j) * 2 + 1] = sb[j];
    }
    if (r == 0) xch[NW * SPG_NP_C * 2 + wave] = sn;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) spg_slots_count_add(p.slots2, SPG_NP_C, (long)p.nblk * 32);
  if (tid < SPG_NP_C) {
    float na = 0.f, mean_ = 0.f, m2_ = 0.f;
    for (int w = 0; w < NW; ++w) {
      const float nb = xch[NW * SPG_NP_C * 2 + w];
      if (nb > 0.f) {
        const float nn = na + nb, f = nb / nn, delta = xch[(w * SPG_NP_C + tid) * 2 + 0] - mean_;
        mean_ += delta * f;
        m2_ += xch[(w * SPG_NP_C + tid) * 2 + 1] + delta * delta * (na * f);
        na = nn;
      }
    }
    if (na > 0.f) spg_slots_add_fwd(p.slots2, SPG_NP_C, tid, na, mean_, m2_);
  }
}

template <int KG>
int launch_pair(const SpgNarrowPairParams& p, hipStream_t stream) {
  constexpr int NW = 8;      // 2 waves per SIMD, each with several blocks: one wave's stores / LDS phase under the other's MFMAs
  const size_t lds = (size_t)(SPG_NP_W2_SLOTS + NW * SPG_NP_A_SLOTS) * sizeof(f32x4) + (size_t)(SPG_NP_C * SPG_NP_W1LD + 2 * SPG_NP_C) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spg_narrow_pair_fwd_kernel<KG, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { spg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int grid = spg_cdiv(p.nblk, NW) < cus ? spg_cdiv(p.nblk, NW) : cus;
  hipLaunchKernelGGL((spg_narrow_pair_fwd_kernel<KG, NW>), dim3(grid), dim3(64 * NW), lds, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}


// ---- backward of the first convolution from ONE pass over the gradient and the cloud (spg_narrow.h) ----
#define SPG_FCB_NW 4      // one wave per SIMD and workgroup (two workgroups fit a CU): a superpoint per wave at a time
// Row <-> operand slot: MFMA (t) of a block takes rows rho(t, h) = 8 (t / 4) + 4 h + (t % 4) from lane half h -- ANY pairing of the
// block's 32 rows with the 16 x 2 reduction slots gives the same sums, and this one lets lane (channel, h) fetch its 16 points as
// four float4 (points 8 q + 4 h .. + 3): coalesced 32-byte runs per channel instead of 16 scattered words.
__device__ __forceinline__ int spg_fcb_row(int t, int h) { return 8 * (t >> 2) + 4 * h + (t & 3); }

// BPS: blocks of 32 points per superpoint (P / 32)
template <bool WANT_DT, int BPS>
__global__ __launch_bounds__(64 * SPG_FCB_NW) void spg_first_conv_bwd_kernel(const SpgFirstConvBwdParams p) {
  constexpr int NW = SPG_FCB_NW, NT = 64 * NW;
  __shared__ double Gs[(SPG_GRAM_MAXF + 1) * (SPG_GRAM_MAXF + 1)];      // workgroup 0 only: Gram matrix of [x; 1]
  __shared__ float cs[4 * SPG_NP_C];                                      // s, c1, mean, d of the 64 channels
  __shared__ float xbar[SPG_GRAM_MAXF + 1];
  __shared__ float vn[2][SPG_GRAM_MAXF + 1];
  __shared__ float k2[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nf = p.nfeat, Cg = nf + 1, npairs = Cg * (Cg + 1) / 2, P = p.P;
  const unsigned long long nrow = p.gram[(size_t)SPG_FOLD_SLOTS * 2 * npairs + 1];
  const double M = nrow != 0ull ? (double)nrow : (double)p.B * (double)P;      // rows behind the Gram matrix (of all ranks under slot-synchronised BatchNorm)

  // ---- prologue: the layer's BatchNorm-backward constants (every workgroup finishes them from the exact sums; workgroup 0 also
  //      writes dgamma / dbeta), the mean of the input from the Gram slots, the small vectors of the data gradient ----
  // (operands that do not depend on the constants are on their way first)
  float w1n[2][2];      // W1[r + 32 j][n]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int n = 0; n < 2; ++n) w1n[j][n] = p.W1[(long)(r + 32 * j) * nf + n];
  if (tid < nf) {       // xbar_k = G[k][last] / M
    const int i = tid, pr = i * Cg - i * (i - 1) / 2 + (nf - i);
    xbar[tid] = (float)(spg_fx_sum<-8>(p.gram + pr, (size_t)npairs, (size_t)2 * npairs) / M);
  }
  spg_bn_fold_bwd(p.fold, blockIdx.x == 0);      // -> p.fold.consts [4][64] in global memory (+ a workgroup barrier)
  for (int i = tid; i < 4 * SPG_NP_C; i += NT) cs[i] = p.fold.consts[i];
  __syncthreads();
  if (blockIdx.x + 1 == gridDim.x) {
    // The LAST workgroup owns no superpoint: it forms the Gram term -d_c sum_j W1[c][j] Gc[j][k] of dW1 -- one more "partial" for the
    // batched reduction -- from the full (centred) Gram matrix while the others are in their main loops (on a working workgroup
    // these ~4 us of float64 were the launch's critical path)
    const bool bad = p.gram[(size_t)SPG_FOLD_SLOTS * 2 * npairs] != 0ull;
    for (int pr = tid; pr < npairs; pr += NT) {
      int i = 0, rem = pr;
      while (rem >= Cg - i) { rem -= Cg - i; ++i; }
      const int j = i + rem;
      double g = spg_fx_sum<-8>(p.gram + pr, (size_t)npairs, (size_t)2 * npairs);
      if (bad) g = __builtin_nan("");
      Gs[i * Cg + j] = g;
      Gs[j * Cg + i] = g;
    }
    __syncthreads();
    float* out = p.partial + (size_t)((int)gridDim.x - 1) * (SPG_NP_C * nf);      // the extra partial
    for (int e = tid; e < SPG_NP_C * nf; e += NT) {
      const int c = e / nf, k = e - c * nf;
      const double xk = Gs[k * Cg + nf] / M;
      double a = 0.0;
#pragma unroll 1
      for (int jj = 0; jj < nf; ++jj) a = fma((double)p.W1[(long)c * nf + jj], Gs[jj * Cg + k] - Gs[jj * Cg + nf] * xk, a);      // Gc[jj][k]
      // (slot-synchronised BatchNorm: G is the Gram matrix of ALL ranks and every rank forms this term -- each contributes its share)
      out[e] = (float)(-(double)cs[3 * SPG_NP_C + c] * a * (p.fold.grad_mul > 0.0 ? p.fold.grad_mul : 1.0));
    }
    return;
  }
  {
    // v_n[k] = sum_c W1[c][k] d_c W1[c][n] (2 nf outputs) and K2_n = sum_c s_c c1_c W1[c][n] (2 outputs): 8 lanes per output
    const int part = tid & 7, nout = 2 * nf + 2;
    for (int o0 = 0; o0 < nout; o0 += NT / 8) {      // (uniform trip count: the exchanges below involve every lane)
      const int o = o0 + (tid >> 3);
      double a = 0.0;
      if (o < nout) {
        const bool isk2 = o >= 2 * nf;
        const int n = isk2 ? o - 2 * nf : o / nf, k = isk2 ? 0 : o - n * nf;
        for (int c = part; c < SPG_NP_C; c += 8) {
          const double wn = (double)p.W1[(long)c * nf + n];
          a = isk2 ? fma((double)cs[c] * (double)cs[SPG_NP_C + c], wn, a) : fma((double)p.W1[(long)c * nf + k] * (double)cs[3 * SPG_NP_C + c], wn, a);
        }
      }
      a += spg_shfl_xor_d(a, 4); a += spg_shfl_xor_d(a, 2); a += spg_shfl_xor_d(a, 1);
      if (o < nout && part == 0) {
        if (o >= 2 * nf) k2[o - 2 * nf] = (float)a; else vn[o / nf][o - (o / nf) * nf] = (float)a;
      }
    }
  }
  __syncthreads();

  // per-lane operands: lane (r, h) is channel r (and r + 32) of the gradient, input channel r (< 16 / 32) of the cloud
  float un[2][2], vl[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    un[n][0] = cs[r] * w1n[0][n];
    un[n][1] = cs[r + 32] * w1n[1][n];
    vl[n] = r < nf ? vn[n][r] : 0.f;
  }
  const float xbl = r < nf ? xbar[r] : 0.f;
  const float k2l = k2[r >> 4];
  const int total = ((int)gridDim.x - 1) * NW;      // (the last workgroup has left)
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  // One superpoint at a time per wave: the loads of ALL its blocks go out first (a wave alone on its SIMD owns 512 registers)
#pragma unroll 1
  for (int g = (int)blockIdx.x * NW + wave; g < p.B; g += total) {
    float gvS[BPS][16][2], xrS[BPS][2];
    f32x4 x4S[BPS][4];
    const float* xc = p.clouds + ((long)g * p.Ctot + (r < nf ? r : 0)) * P + 4 * h;
#pragma unroll
    for (int b = 0; b < BPS; ++b) {
      const float* gb = p.g + ((long)g * P + 32 * b) * SPG_NP_C + r;
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) gvS[b][t][j] = gb[spg_fcb_row(t, h) * SPG_NP_C + 32 * j];
#pragma unroll
      for (int q = 0; q < 4; ++q) x4S[b][q] = *reinterpret_cast<const f32x4*>(xc + 32 * b + 8 * q);
      xrS[b][0] = 0.f; xrS[b][1] = 0.f;
      if (WANT_DT) {      // raw x / y of the row this lane holds after the butterfly: entry t = r & 15
        const float* rb = p.clouds + ((long)g * p.Ctot) * P + 32 * b + spg_fcb_row(r & 15, h);
        xrS[b][0] = rb[0]; xrS[b][1] = rb[P];
      }
    }
    f32x4 T = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.stnT != nullptr) T = *reinterpret_cast<const f32x4*>(p.stnT + (long)g * 4);
    float dta0 = 0.f, dta1 = 0.f;
#pragma unroll
    for (int b = 0; b < BPS; ++b) {
      float (&gv)[16][2] = gvS[b];
      float xa[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float v = x4S[b][t >> 2][t & 3];
        if (p.stnT != nullptr) {      // channel 0 (x) and channel 1 (y) sit in neighbouring lanes: learning/pointnet.py:123, as spg_fetch
          const float other = spg_lane_xor1(v);
          const float tx = fmaf(v, T[0] + 1.f, other * T[2]), ty = fmaf(other, T[1], v * (T[3] + 1.f));      // (branch-free: selects)
          v = r == 0 ? tx : (r == 1 ? ty : v);
        }
        xa[t] = r < nf ? v - xbl : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t], gv[t][j], acc[j], 0, 0, 0);
      if (WANT_DT) {
        // V[t]: this lane's share of dxy[row rho(t, h)][n].  A transposing butterfly over lane bits 0-3 leaves lane r with the sum
        // over its 16-lane group of entry t = r & 15 (15 exchanges instead of 16 x 4), one more exchange completes the half-wave;
        // lanes r < 16 keep n = 0, the others n = 1
        float dxy = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          float V[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) V[t] = fmaf(gv[t][0], un[n][0], fmaf(gv[t][1], un[n][1], -(xa[t] * vl[n])));
#pragma unroll
          for (int m = 8; m >= 1; m >>= 1) {
            const bool up = (r & m) != 0;
#pragma unroll
            for (int i = 0; i < m; ++i) {
              // (the two values as opaque registers: left visible, the optimiser turns `up ? V[i] : V[i + m]` into ONE load with a
              //  lane-dependent index into V -- a 16-way select chain per element, 848 spilled SGPRs)
              float lo_ = V[i], hi_ = V[i + m];
              asm volatile("" : "+v"(lo_), "+v"(hi_));
              const float send = up ? lo_ : hi_, keep = up ? hi_ : lo_;
              V[i] = keep + __shfl_xor(send, m, 64);
            }
          }
          const float tot = V[0] + __shfl_xor(V[0], 16, 64);
          if ((r >> 4) == n) dxy = tot - k2l;      // entry: n = r >> 4, row rho(r & 15, h)
        }
        dta0 = fmaf(xrS[b][0], dxy, dta0);
        dta1 = fmaf(xrS[b][1], dxy, dta1);
      }
    }
    if (WANT_DT) {      // dT[g][2 a + n] = sum over the lanes that hold entry n
      const float a00 = spg_wave_sum(r < 16 ? dta0 : 0.f), a01 = spg_wave_sum(r < 16 ? 0.f : dta0);
      const float a10 = spg_wave_sum(r < 16 ? dta1 : 0.f), a11 = spg_wave_sum(r < 16 ? 0.f : dta1);
      if (lane == 0) { float* o = p.dT + (long)g * 4; o[0] = a00; o[1] = a01; o[2] = a10; o[3] = a11; }
    }
  }
  // ---- the workgroup's partial of dW1 (accumulator rows = input channels): the four waves' accumulators meet in LDS and are added in
  //      wave order by all threads (one partial per WAVE made the step's batched reduction 4 us longer than this exchange costs);
  //      s_c sum_rows g x'' ----
  {
    extern __shared__ float red[];      // [NW][32][64]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) red[(wave * 32 + j * 16 + q) * 64 + lane] = acc[j][q];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * (SPG_NP_C * nf);
    for (int e = tid; e < 32 * 64; e += NT) {
      const int jq = e >> 6, l = e & 63, j = jq >> 4, q = jq & 15;
      const int k = spg_acc_row(q, l >> 5), c = (l & 31) + 32 * j;
      if (k < nf) {
        float a = red[e];
#pragma unroll
        for (int w = 1; w < NW; ++w) a += red[w * 32 * 64 + e];
        out[c * nf + k] = cs[c] * a;
      }
    }
  }
}

}  // namespace

bool spg_narrow_pair_supported(int nfeat, int c1, int c2, int P, long M) {
  return nfeat >= 1 && nfeat <= SPG_GRAM_MAXF && c1 == SPG_NP_C && c2 == SPG_NP_C && P >= 32 && P <= 128 && P % 32 == 0 && M > 0 &&
         M * SPG_NP_C * 4 < (1L << 32) && M * (long)nfeat * 4 < (1L << 32) && spg_tune_get(SPG_TUNE_NO_NARROW_PAIR) != 1;
}

int spg_launch_cloud_gram(const SpgGramParams& p, hipStream_t stream) {
  SPG_CHECK_ARG(p.clouds && p.gram && p.B > 0 && p.P >= 1 && p.P <= 128 && p.nfeat >= 1 && p.nfeat <= SPG_GRAM_MAXF, "gram arguments");
  if (p.nfeat + 1 <= 16 && p.P % 16 == 0 && spg_tune_get(SPG_TUNE_NO_NARROW_PAIR) != 2)      // (key 17 = 2: the general Gram kernel, for the equality test)
    hipLaunchKernelGGL(spg_cloud_gram16_kernel, dim3(spg_cdiv(p.B, 4 * SPG_GRAM16_SPW)), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(spg_cloud_gram_kernel, dim3(p.B), dim3(256), 0, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

int spg_launch_narrow_pair_fwd(const SpgNarrowPairParams& p, hipStream_t stream) {
  SPG_CHECK_ARG(p.clouds && p.W1 && p.W2 && p.y1 && p.y2 && p.gram && p.slots2 && p.mean1 && p.rstd1 && p.s1 && p.t1, "null pointer");
  SPG_CHECK_ARG(p.P % 32 == 0 && p.nblk > 0 && p.count > 0.0, "blocks of 32 points");
  // (counted as what it replaces: the two forward GEMMs 2 M 64 nfeat + 2 M 64 64; tag kind 5 = one-pass narrow layers)
  SpgProfSpan prof(stream, 2.0 * p.count * SPG_NP_C * (p.nfeat + SPG_NP_C), 5000000 + 1, SPG_NP_C, SPG_NP_C);
  return p.nfeat <= 16 ? launch_pair<2>(p, stream) : launch_pair<4>(p, stream);
}

bool spg_first_conv_bwd_supported(int nfeat, int c1, int P, long M) {
  return nfeat >= 2 && nfeat <= SPG_GRAM_MAXF && c1 == SPG_NP_C && P >= 32 && P <= 128 && P % 32 == 0 && M > 0 && M * SPG_NP_C * 4 < (1L << 32) &&
         spg_tune_get(SPG_TUNE_NO_NARROW_PAIR) == 0 && !spg_tune_get(SPG_TUNE_NO_FIRST_CONV_BWD);
}

static int fcb_grid(int B) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int need = spg_cdiv(B, SPG_FCB_NW);
  return need < 2 * cus ? need : 2 * cus;
}
// partials of dW1 the kernel writes: one per workgroup + the Gram term
int spg_first_conv_bwd_partials(int B) { return fcb_grid(B) + 1; }

int spg_launch_first_conv_bwd(const SpgFirstConvBwdParams& p, hipStream_t stream) {
  SPG_CHECK_ARG(p.clouds && p.g && p.W1 && p.gram && p.partial && p.fold.slots && p.fold.consts && p.fold.C == SPG_NP_C, "first-convolution backward arguments");
  const int grid = fcb_grid(p.B);
  // (counted as what it replaces: the weight gradient 2 M 64 nfeat and, with dT, the two-column data gradient 2 M 2 64)
  SpgProfSpan prof(stream, 2.0 * (double)p.B * p.P * SPG_NP_C * (p.nfeat + (p.dT != nullptr ? 2 : 0)), 5000000 + 2, SPG_NP_C, p.nfeat);
  const size_t lds = (size_t)SPG_FCB_NW * 32 * 64 * sizeof(float);
  const dim3 gd(grid + 1), bd(64 * SPG_FCB_NW);      // (+ the workgroup of the Gram term)
#define SPG_FCB_LAUNCH(DT, BPS_) hipLaunchKernelGGL((spg_first_conv_bwd_kernel<DT, BPS_>), gd, bd, lds, stream, p)
  const int bps = p.P / 32;
  if (p.dT != nullptr) { if (bps == 1) SPG_FCB_LAUNCH(true, 1); else if (bps == 2) SPG_FCB_LAUNCH(true, 2); else if (bps == 3) SPG_FCB_LAUNCH(true, 3); else SPG_FCB_LAUNCH(true, 4); }
  else { if (bps == 1) SPG_FCB_LAUNCH(false, 1); else if (bps == 2) SPG_FCB_LAUNCH(false, 2); else if (bps == 3) SPG_FCB_LAUNCH(false, 3); else SPG_FCB_LAUNCH(false, 4); }
#undef SPG_FCB_LAUNCH
  SPG_LAUNCH_CHECK();
  return 0;
}
