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
