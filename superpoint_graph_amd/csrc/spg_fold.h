// Fixed-point BatchNorm statistics slots: device-side helpers shared by the GEMM kernels (spg_gemm.hip) and the narrow-layer
// kernels (spg_narrow.hip).  Layout and rationale: spg_gemm.h (SpgBnFold).  Device code only.
#pragma once
#include "spg_gemm.h"

// ---- fixed-point statistics slots (SpgBnFold, spg_gemm.h) ----
// A double v is split as v * 2^SH = hi + lo * 2^-44 (hi = floor, lo in [0, 2^44)); each limb is an exact int64 sum, 2^19
// contributions per slot without overflow, |hi| <= 2^44 per contribution.  Two scalings:
//   forward sums (sum x, sum x^2), SH = -8: |v| <= 2^52 per contribution, quantum 2^-36 -- a persistent workgroup's partial
//     over 512 rows stays in range up to a pre-BatchNorm rms of ~3e6 (un-normalised metre coordinates with pc_xyznormalize 0;
//     the 2^36 range of round 3 turned rms > 1e4 into NaN statistics where the reference's fp32 stays finite);
//     a quantum of 1.5e-11 per contribution is invisible next to eps = 1e-5 in var + eps and to fp32 in the mean;
//   backward sums (sum dz, sum dz * xhat), SH = +8: gradients are SMALL numbers -- quantum 2^-52, |v| <= 2^36.
template <int SH>
__device__ __forceinline__ void spg_fx_split(double v, long long& hi, long long& lo) {
  constexpr double LIM = SH < 0 ? 0x1p52 : 0x1p36, SC = SH < 0 ? 0x1p-8 : 0x1p8;
  v = fmin(fmax(v, -LIM), LIM);
  const double t = v * SC, f = floor(t);
  hi = (long long)f;                        // |hi| <= 2^44
  lo = (long long)((t - f) * 0x1p44);       // [0, 2^44): 2^19 contributions fit one int64 slot
}
// The consumer's side: the SPG_FOLD_SLOTS slots of one sum are added EXACTLY (128-bit integers: 8 x 2^63 * 2^44 fits) and
// converted once -- the result does not depend on which workgroup used which slot, i.e. not on the launch geometry (a
// grouped launch numbers its workgroups differently from the stand-alone launch of the same job; both give the same bits).
// slots: first limb of slot 0 (hi); lo limb at +C; next slot at +stride
template <int SH>
__device__ __forceinline__ double spg_fx_sum(const unsigned long long* __restrict__ s, size_t C, size_t stride) {
  __int128 t = 0;
#pragma unroll
  for (int k = 0; k < SPG_FOLD_SLOTS; ++k) {
    const long long hi = (long long)s[k * stride], lo = (long long)s[k * stride + C];
    t += ((__int128)hi << 44) + (__int128)lo;
  }
  // sign and magnitude: both halves of |t| are non-negative, so the two conversions cannot cancel (two's-complement halves of a
  // small negative total would: -2^64 + (2^64 - x rounded to 53 bits))
  const bool neg = t < 0;
  const unsigned __int128 u = neg ? (unsigned __int128)(-t) : (unsigned __int128)t;
  const double mag = (double)(unsigned long long)(u >> 64) * 0x1p64 + (double)(unsigned long long)u;
  constexpr double ISC = SH < 0 ? 0x1p8 : 0x1p-8;
  return (neg ? -mag : mag) * (0x1p-44 * ISC);
}

// one contribution (two sums) of column `col` into the layer's slots
template <int SH>
__device__ __forceinline__ void spg_slots_add_t(unsigned long long* slots, int C, int col, double sx, double sxx) {
  constexpr double LIM = SH < 0 ? 0x1p52 : 0x1p36;
  unsigned long long* s = slots + (size_t)(blockIdx.x & (SPG_FOLD_SLOTS - 1)) * 4 * C + col;
  if (!(fabs(sx) <= LIM && fabs(sxx) <= LIM)) atomicOr(slots + (size_t)SPG_FOLD_SLOTS * 4 * C, 1ull);      // NaN / inf / out of range
  long long hi, lo;
  spg_fx_split<SH>(sx, hi, lo);
  __hip_atomic_fetch_add(s, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(s + C, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  spg_fx_split<SH>(sxx, hi, lo);
  __hip_atomic_fetch_add(s + 2 * (size_t)C, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(s + 3 * (size_t)C, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// backward: (sum dz, sum dz * xhat)
__device__ __forceinline__ void spg_slots_add(unsigned long long* slots, int C, int col, double sx, double sxx) {
  spg_slots_add_t<8>(slots, C, col, sx, sxx);
}
// forward: (rows n, mean, M2 of those rows) -> (sum x, sum x^2)
__device__ __forceinline__ void spg_slots_add_fwd(unsigned long long* slots, int C, int col, float n, float mean, float m2) {
  spg_slots_add_t<-8>(slots, C, col, (double)n * (double)mean, (double)m2 + (double)n * (double)mean * (double)mean);
}


// The rows behind a layer's sums travel WITH them: the spare word behind the flag word counts them (every producer launch adds
// its rows once).  Under slot-synchronised BatchNorm the all-reduce of the slots therefore delivers the rows of ALL ranks to the
// consumer -- no second collective, no host round trip; a single rank reads back its own count.  0 (a producer that does not
// count): the caller's value.
__device__ __forceinline__ void spg_slots_count_add(unsigned long long* slots, int C, long rows) {
  __hip_atomic_fetch_add(slots + (size_t)SPG_FOLD_SLOTS * 4 * C + 1, (unsigned long long)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double spg_slots_count(const unsigned long long* slots, int C, double fallback) {
  const unsigned long long n = slots[(size_t)SPG_FOLD_SLOTS * 4 * C + 1];
  return n != 0ull ? (double)n : fallback;
}

// backward consumer prologue (weight-gradient kernels): sums (sum dz, sum dz * xhat) of the layer whose BatchNorm-backward
// formula this launch's `a` operand applies -> consts [4][C] = {s, c1, mean, s * c2 * rstd}; workgroup 0 also writes the
// BatchNorm parameter gradients.  All threads of the workgroup; ends with a workgroup barrier.
// `first`: exactly one workgroup of the step passes true (it writes dgamma / dbeta)
__device__ __forceinline__ void spg_bn_fold_bwd(const SpgBnFoldBwd& f, const bool first) {
  const int C = f.C;
  const bool bad = f.slots[(size_t)SPG_FOLD_SLOTS * 4 * C] != 0ull;
  const double cnt = spg_slots_count(f.slots, C, f.count);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double a = spg_fx_sum<8>(f.slots + c, (size_t)C, (size_t)4 * C);
    const double b = spg_fx_sum<8>(f.slots + 2 * (size_t)C + c, (size_t)C, (size_t)4 * C);
    if (bad) a = __builtin_nan("");
    const float ps = f.s[c], pmean = f.mean[c], prstd = f.rstd[c];
    const double c1 = a / cnt, c2 = b / cnt;
    f.consts[0 * C + c] = ps;
    f.consts[1 * C + c] = (float)c1;
    f.consts[2 * C + c] = pmean;
    f.consts[3 * C + c] = (float)((double)ps * c2 * (double)prstd);
    if (first) {
      const double gm = f.grad_mul > 0.0 ? f.grad_mul : 1.0;
      if (f.dbeta) f.dbeta[c] = (float)(a * gm);
      if (f.dgamma) f.dgamma[c] = (float)(b * gm);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// consumer prologue: all threads of the workgroup; ends with a workgroup barrier behind which s / t / mean / rstd are readable
// `first`: exactly one workgroup of the launch passes true (it advances the running statistics)
__device__ __forceinline__ void spg_bn_fold_fwd(const SpgBnFold& f, const bool first) {
  const int C = f.C;
  const bool bad = f.slots[(size_t)SPG_FOLD_SLOTS * 4 * C] != 0ull;
  const double M = spg_slots_count(f.slots, C, f.count);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    // every slot is an exact integer sum; the slots are added exactly too (spg_fx_sum)
    double sx = spg_fx_sum<-8>(f.slots + c, (size_t)C, (size_t)4 * C);
    const double sxx = spg_fx_sum<-8>(f.slots + 2 * (size_t)C + c, (size_t)C, (size_t)4 * C);
    if (bad) sx = __builtin_nan("");
    const double mean = sx / M;
    double m2 = sxx - M * mean * mean;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / M;
    const double rstd = 1.0 / sqrt(var + (double)f.eps);
    const double g = f.gamma ? (double)f.gamma[c] : 1.0, be = f.beta ? (double)f.beta[c] : 0.0;
    f.mean[c] = (float)mean;
    f.rstd[c] = (float)rstd;
    f.s[c] = (float)(g * rstd);
    f.t[c] = (float)(be - mean * g * rstd);
    if (first && f.rm != nullptr && f.update_times > 0) {
      const double uvar = M > 1.0 ? m2 / (M - 1.0) : var;
      float rm = f.rm[c], rv = f.rv[c];
      for (int u = 0; u < f.update_times; ++u) {
        rm = (1.f - f.momentum) * rm + f.momentum * (float)mean;
        rv = (1.f - f.momentum) * rv + f.momentum * (float)uvar;
      }
      f.rm[c] = rm;
      f.rv[c] = rv;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's copies of s / t have left for L2 ...
  __syncthreads();                                         // ... before any of its waves loads them
}

