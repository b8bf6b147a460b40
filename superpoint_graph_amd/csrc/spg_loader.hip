// Device-side superpoint loader: the per-superpoint body of load_superpoint (reference learning/spg.py:198-236)
// and the geometric part of augment_cloud (:239-258) over a RAGGED point buffer.
//
// The reference reads every superpoint's points from HDF5, resamples them to ptn_npts rows with numpy, normalises xyz,
// selects the feature columns and transposes -- one python call per superpoint, in DataLoader workers.  Here the raw
// points of a whole scene (or batch) sit in HBM once as points[Ntot][ncols] + offsets[S+1]; the host keeps only the
// random streams (the resampling indices `rs.choice`, the augmentation matrix, optionally the jitter noise), so the
// numpy RNG sequence -- and therefore test-time parity -- is unchanged.  One workgroup per superpoint:
//   gather the sampled rows -> xyz min / max / mean (fp32, numpy's row-sequential summation order) ->
//   (xyz - mean) / (diameter + 1e-10) -> feature selection -> [x y z] M^T in fp64 -> + jitter -> write [F][npts].
// HBM-bound integer/byte-style work: algorithmic traffic = npts * (ncols + F) * 4 B per superpoint; rows are
// gathered as whole 56/60-byte records (one or two 64-B sectors each), the output is written coalesced along points.
#include "../../include/spg_hip.h"
#include "spg_common.h"
#include <float.h>

namespace {

struct LoaderParams {
  const float* points;
  int ncols;
  const int64_t* offsets;
  const int* slot;
  const int* sample_idx;
  int npts, xyznormalize, nfeat;
  int colmap[SPG_LOADER_MAX_FEATS];
  const double* M;
  const float* noise;
  float* clouds;
  float* diam;
};

__global__ __launch_bounds__(256) void spg_load_superpoints_kernel(const LoaderParams p) {
  extern __shared__ float lds[];                 // xyz[3][npts] | red[6][waves]
  const int s = blockIdx.x;
  const int slot = p.slot[s];
  if (slot < 0) return;                          // too few points: no cloud (flag -1 on the host side)
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  float* sx = lds;
  float* red = lds + 3 * p.npts;
  const float* base = p.points + p.offsets[s] * (long)p.ncols;
  const int* idx = p.sample_idx + (long)s * p.npts;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int q = tid; q < p.npts; q += nthr) {
    const float* row = base + (long)idx[q] * p.ncols;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = row[j];
      sx[j * p.npts + q] = v;
      mn[j] = fminf(mn[j], v);
      mx[j] = fmaxf(mx[j], v);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[j] = fminf(mn[j], __shfl_xor(mn[j], off, 64));
      mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], off, 64));
    }
    if (lane == 0) { red[j * nw + wave] = mn[j]; red[(3 + j) * nw + wave] = mx[j]; }
  }
  __syncthreads();
  // mean: one thread per coordinate adds the npts values in row order, exactly like numpy's float32 add.reduce
  // over axis 0 of the [npts, 3] view (spg.py:219) -- the division by npts follows
  __shared__ float s_mean[3], s_diam;
  if (tid < 3) {
    float acc = 0.f;
    const float* v = sx + tid * p.npts;
    for (int q = 0; q < p.npts; ++q) acc += v[q];
    s_mean[tid] = acc / (float)p.npts;
  }
  if (tid == 64 || (nthr == 64 && tid == 3)) {
    float d = -FLT_MAX;
    for (int j = 0; j < 3; ++j) {
      float a = FLT_MAX, b = -FLT_MAX;
      for (int w = 0; w < nw; ++w) { a = fminf(a, red[j * nw + w]); b = fmaxf(b, red[(3 + j) * nw + w]); }
      d = fmaxf(d, b - a);
    }
    s_diam = d;
  }
  __syncthreads();
  const float diameter = p.xyznormalize ? s_diam : 0.f;
  const float denom = s_diam + 1e-10f;           // float32 + python float stays float32 (spg.py:219)
  if (tid == 0) p.diam[slot] = diameter;
  float* out = p.clouds + (long)slot * p.nfeat * p.npts;
  const double* M = p.M ? p.M + (long)s * 9 : nullptr;
  for (int q = tid; q < p.npts; q += nthr) {
    const float* row = base + (long)idx[q] * p.ncols;
    float xyz[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float c = sx[j * p.npts + q] - s_mean[j];
      xyz[j] = p.xyznormalize ? c / denom : c;
    }
    float f[SPG_LOADER_MAX_FEATS];
#pragma unroll
    for (int k = 0; k < SPG_LOADER_MAX_FEATS; ++k) {
      if (k < p.nfeat) {
        const int c = p.colmap[k];
        f[k] = c < 3 ? xyz[c] : row[c];
      }
    }
    if (M != nullptr && p.nfeat >= 3) {          // P[:, :3] = np.dot(P[:, :3], M.T): float64 product, rounded on the store
      const double a = (double)f[0], b = (double)f[1], c = (double)f[2];
      const float r0 = (float)(a * M[0] + b * M[1] + c * M[2]);
      const float r1 = (float)(a * M[3] + b * M[4] + c * M[5]);
      const float r2 = (float)(a * M[6] + b * M[7] + c * M[8]);
      f[0] = r0; f[1] = r1; f[2] = r2;
    }
    const float* nz = p.noise ? p.noise + ((long)slot * p.npts + q) * p.nfeat : nullptr;
#pragma unroll
    for (int k = 0; k < SPG_LOADER_MAX_FEATS; ++k) {
      if (k < p.nfeat) {
        float v = f[k];
        if (nz != nullptr) v += nz[k];           // jitter (spg.py:255-257), drawn by the host
        out[(long)k * p.npts + q] = v;           // cloud.T: channel-major, coalesced along the points
      }
    }
  }
}

}  // namespace

extern "C" int spg_load_superpoints(const float* points, int ncols, const int64_t* offsets, int n_superpoints,
                                    const int32_t* slot, const int32_t* sample_idx, int npts, int xyznormalize,
                                    const int32_t* colmap, int nfeat, const double* M, const float* noise, float* clouds,
                                    float* diam, void* stream) {
  SPG_CHECK_ARG(points && offsets && slot && sample_idx && colmap && clouds && diam, "null pointer");
  SPG_CHECK_ARG(n_superpoints >= 0 && npts >= 1 && npts <= 4096, "npts must be in [1, 4096]");
  SPG_CHECK_ARG(ncols >= 3 && nfeat >= 1 && nfeat <= SPG_LOADER_MAX_FEATS, "ncols >= 3, 1 <= nfeat <= SPG_LOADER_MAX_FEATS");
  if (n_superpoints == 0) return 0;
  LoaderParams p;
  p.points = points; p.ncols = ncols; p.offsets = offsets; p.slot = slot; p.sample_idx = sample_idx; p.npts = npts;
  p.xyznormalize = xyznormalize; p.nfeat = nfeat; p.M = M; p.noise = noise; p.clouds = clouds; p.diam = diam;
  for (int k = 0; k < SPG_LOADER_MAX_FEATS; ++k) p.colmap[k] = 0;
  for (int k = 0; k < nfeat; ++k) {
    SPG_CHECK_ARG(colmap[k] >= 0 && colmap[k] < ncols, "colmap entry outside the raw columns");
    p.colmap[k] = colmap[k];
  }
  const int threads = npts <= 64 ? 64 : (npts <= 128 ? 128 : 256);
  const size_t shmem = ((size_t)3 * npts + 6 * (threads / 64)) * sizeof(float);
  hipLaunchKernelGGL(spg_load_superpoints_kernel, dim3(n_superpoints), dim3(threads), shmem, (hipStream_t)stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Device-side random streams of the loader (optional mode, `--loader_rng device`): the resampling indices, the
// augmentation matrix and the jitter of every superpoint from a counter-based generator (Philox4x32-10, Salmon et al.
// SC'11) keyed by (seed, superpoint id, step) -- no host loop over the superpoints, reproducible for a given seed, and
// independent of the batch composition.  It is a DIFFERENT stream from numpy's MT19937 (statistically equivalent draws:
// uniform indices with replacement, uniform angle / scale, Bernoulli mirrors, clipped N(0, 0.01^2) jitter, spg.py:207-258);
// the default host-stream mode reproduces the reference's clouds bit for bit.
// ---------------------------------------------------------------------------------------------
namespace {
struct U4 { unsigned x, y, z, w; };
__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    U4 n;
    n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0; n.y = (unsigned)p1;
    n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1; n.w = (unsigned)p0;
    c = n;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}

struct RngParams {
  const int64_t* counts;      // [S] points per superpoint
  const int64_t* ids;         // [S] superpoint ids (key of the stream)
  const int* slot;            // [S] row in the cloud tensor or -1
  int S, npts, nfeat;
  unsigned seed_lo, seed_hi, step;
  float scale, mirror_prob;
  int rot, jitter, augment;
  int* sidx; double* M; float* noise;
};

// kinds (counter word 1): 0 indices (element = point / 4), 1 augmentation scalars, 2 jitter (element = value / 4)
__global__ void spg_loader_rng_kernel(const RngParams p) {
  const int s = blockIdx.x;
  const long n = (long)p.counts[s];
  const unsigned idl = (unsigned)p.ids[s], idh = (unsigned)((unsigned long long)p.ids[s] >> 32) ^ p.step;
  for (int q4 = threadIdx.x; 4 * q4 < p.npts; q4 += blockDim.x) {
    const U4 u = philox4x32_10(U4{(unsigned)q4, 0u, idl, idh}, p.seed_lo, p.seed_hi);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    for (int e = 0; e < 4 && 4 * q4 + e < p.npts; ++e) {
      const int q = 4 * q4 + e;
      // n > npts: npts draws with replacement; n < npts: all points, then npts - n draws; (uniform in [0, n): high word of u * n)
      const int v = (n <= p.npts && q < n) ? q : (int)(((unsigned long long)w[e] * (unsigned long long)(n > 0 ? n : 1)) >> 32);
      p.sidx[(long)s * p.npts + q] = v;
    }
  }
  if (p.augment && p.M != nullptr && threadIdx.x == 0) {
    const U4 u = philox4x32_10(U4{0u, 1u, idl, idh}, p.seed_lo, p.seed_hi);
    const double two32 = 4294967296.0;
    double sc = 1.0;
    if (p.scale > 1.f) sc = 1.0 / (double)p.scale + ((double)u.x + 0.5) / two32 * ((double)p.scale - 1.0 / (double)p.scale);
    double c = 1.0, sn = 0.0;
    if (p.rot) { const double a = ((double)u.y + 0.5) / two32 * 6.283185307179586; c = cos(a); sn = sin(a); }
    double mx = 1.0, my = 1.0;
    if (p.mirror_prob > 0.f) {
      if (((double)u.z + 0.5) / two32 < 0.5 * (double)p.mirror_prob) mx = -1.0;
      if (((double)u.w + 0.5) / two32 < 0.5 * (double)p.mirror_prob) my = -1.0;
    }
    // M = mirror_y * mirror_x * Rz(a) * (sc I)   (composition order of augment_cloud, spg.py:241-251)
    double* M = p.M + (long)s * 9;
    M[0] = mx * c * sc;  M[1] = -mx * sn * sc; M[2] = 0.0;
    M[3] = my * sn * sc; M[4] = my * c * sc;   M[5] = 0.0;
    M[6] = 0.0;          M[7] = 0.0;           M[8] = sc;
  }
  const int slot = p.slot[s];
  if (p.jitter && p.noise != nullptr && slot >= 0) {
    const int total = p.npts * p.nfeat;
    float* out = p.noise + (long)slot * total;
    for (int q4 = threadIdx.x; 4 * q4 < total; q4 += blockDim.x) {
      const U4 u = philox4x32_10(U4{(unsigned)q4, 2u, idl, idh}, p.seed_lo, p.seed_hi);
      // two Box-Muller pairs per block: sigma 0.01, clipped to +-0.05 (spg.py:255-257)
      const float u1 = ((float)(u.x >> 8) + 0.5f) * 5.9604645e-8f, u2 = ((float)(u.y >> 8) + 0.5f) * 5.9604645e-8f;
      const float u3 = ((float)(u.z >> 8) + 0.5f) * 5.9604645e-8f, u4 = ((float)(u.w >> 8) + 0.5f) * 5.9604645e-8f;
      const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
      const float z[4] = {r1 * cosf(6.2831853f * u2), r1 * sinf(6.2831853f * u2), r2 * cosf(6.2831853f * u4), r2 * sinf(6.2831853f * u4)};
      for (int e = 0; e < 4 && 4 * q4 + e < total; ++e) out[4 * q4 + e] = fminf(fmaxf(0.01f * z[e], -0.05f), 0.05f);
    }
  }
}
}  // namespace

extern "C" int spg_loader_random(const int64_t* counts, const int64_t* ids, const int32_t* slot, int n_superpoints, int npts,
                                 int nfeat, uint64_t seed, uint32_t step, int augment, float scale, int rot, float mirror_prob,
                                 int jitter, int32_t* sample_idx, double* M, float* noise, void* stream) {
  SPG_CHECK_ARG(counts && ids && slot && sample_idx && npts >= 1 && nfeat >= 1, "bad argument");
  if (n_superpoints == 0) return 0;
  RngParams p;
  p.counts = counts; p.ids = ids; p.slot = slot; p.S = n_superpoints; p.npts = npts; p.nfeat = nfeat;
  p.seed_lo = (unsigned)seed; p.seed_hi = (unsigned)(seed >> 32); p.step = step;
  p.scale = scale; p.mirror_prob = mirror_prob; p.rot = rot; p.jitter = jitter; p.augment = augment;
  p.sidx = sample_idx; p.M = M; p.noise = noise;
  hipLaunchKernelGGL(spg_loader_rng_kernel, dim3(n_superpoints), dim3(128), 0, (hipStream_t)stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Evaluation accounting on the device (reference learning/main.py:246-263 and eval_final :267-311 with
// learning/metrics.py:16-18): mean of the logits over the test-time samples, arg-max prediction of every superpoint,
// and for the superpoints with ground truth (label_mode != -100, main.py:447-452) the confusion-matrix update
// confusion[:, pred_i] += label_vec[i, :] plus the top-1 accuracy counters.  Integer accumulation with 64-bit atomics:
// exact, hence order-independent.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void spg_eval_accumulate_kernel(const float* __restrict__ logits, int S, long sstride, int N, int C,
                                           const int64_t* __restrict__ label_mode, const int64_t* __restrict__ label_vec,
                                           int64_t* __restrict__ pred, unsigned long long* __restrict__ confusion,
                                           unsigned long long* __restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int best = 0;
  float bv = 0.f;
  for (int c = 0; c < C; ++c) {
    float v = logits[(long)i * C + c];
    // np.mean(np.stack(o, 0), 0): float32 sum over the samples in order, then one division (main.py:296-297)
    for (int s = 1; s < S; ++s) v += logits[s * sstride + (long)i * C + c];
    if (S > 1) v = v / (float)S;
    if (c == 0 || v > bv) { bv = v; best = c; }      // np.argmax: the first maximum
  }
  pred[i] = best;
  const int64_t t = label_mode[i];
  if (t == -100) return;                              // no ground truth: not counted (filter_valid)
  for (int c = 0; c < C; ++c) {
    const int64_t n = label_vec[(long)i * C + c];
    if (n != 0) atomicAdd(&confusion[(long)c * C + best], (unsigned long long)n);
  }
  atomicAdd(&counters[1], 1ull);
  if ((int64_t)best == t) atomicAdd(&counters[0], 1ull);
}
}  // namespace

extern "C" int spg_eval_accumulate(const float* logits, int n_samples, long sample_stride, int N, int C,
                                   const int64_t* label_mode, const int64_t* label_vec, int64_t* pred,
                                   int64_t* confusion, int64_t* counters, void* stream) {
  SPG_CHECK_ARG(logits && label_mode && label_vec && pred && confusion && counters, "null pointer");
  SPG_CHECK_ARG(n_samples >= 1 && N >= 0 && C >= 1, "n_samples >= 1, C >= 1");
  if (N == 0) return 0;
  hipLaunchKernelGGL(spg_eval_accumulate_kernel, dim3(spg_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, logits, n_samples,
                     sample_stride, N, C, label_mode, label_vec, pred, reinterpret_cast<unsigned long long*>(confusion),
                     reinterpret_cast<unsigned long long*>(counters));
  SPG_LAUNCH_CHECK();
  return 0;
}
