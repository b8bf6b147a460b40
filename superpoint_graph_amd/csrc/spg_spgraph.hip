// Superpoint-graph construction on the device (SURVEY.md section 8, row f4 tail: partition/graphs.py:75-210 compute_sp_graph
// after the Delaunay triangulation, and partition/ply_c/ply_c.cpp:384-462 compute_geof).
//
// compute_sp_graph: the reference walks every tetrahedron of scipy's Delaunay triangulation, keeps the vertex pairs that join
// two different components (superpoints), makes them unique (np.unique over 2 x 12T columns), drops those longer than d_max,
// orders them by (component of source, component of target) and then loops in Python over the components (covariance
// eigenvalues of the unique points) and over the superedges (offset statistics).  Here:
//   spg_spg_tet_edges    one thread per (tetrahedron, vertex pair): both directions of every interface pair as a 64-bit key
//                        (source << 32 | target), appended with one atomic per wavefront;
//   spg_spg_unique_edges radix sort of the keys (rocPRIM), first-of-run + length flags -- the float32 distance is evaluated
//                        in the reference's operation order (graphs.py:113, no fused multiply-add), so the SET of surviving edges
//                        is bit-exact --, stable compaction, component-pair key per edge;
//   spg_spg_group_edges  stable sort by the component-pair key (graphs.py:121-125; inside a group the edges stay in (source,
//                        target) order -- the reference's argsort leaves that order unspecified), run-length encoding -> superedges;
//   spg_spg_superpoints  points ordered by (component, x, y, z) with two stable 64-bit sorts, one wavefront per component:
//                        duplicates dropped (np.unique(xyz[comp], axis=0), :150), mean and covariance of the unique points in
//                        float64, eigenvalues by cyclic Jacobi sweeps (np.cov + LA.eig, :162), label histogram by integer atomics;
//   spg_spg_superedges   one wavefront per superedge: mean / std / mean norm of the offsets in float64, ratios in the
//                        reference's own float32 / float64 expressions (:186-201).
// Integer outputs (source, target, point counts, label histograms, the edge set) are bit-exact; the float features are
// accumulated in float64 and rounded once (the reference accumulates in float32: tolerances in tests/test_gpu_spgraph.py).
// HBM-bound integer / byte work; nothing here is shaped into a GEMM.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "../../include/spg_hip.h"
#include "spg_common.h"

namespace {

typedef unsigned long long u64;

#define SPG_RP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) {                                                              \
      spg_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));  \
      return (int)e__;                                                                    \
    }                                                                                     \
  } while (0)

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- float -> unsigned with the same order; -0.0 and +0.0 are one value (np.unique compares with ==) ----
__device__ __forceinline__ unsigned ordered_bits(float f) {
  unsigned b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// -------------------------------------------------------------------------------------------------------------------
// interface edges of the tetrahedra (graphs.py:85-107)
// -------------------------------------------------------------------------------------------------------------------
__global__ void tet_edges_kernel(const int32_t* __restrict__ tets, long T, const int32_t* __restrict__ comp, u64* __restrict__ keys,
                                 long capacity, u64* __restrict__ counter) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long t = gid / 6;
  const int pr = (int)(gid - t * 6);
  // pairs in the reference's order: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
  const int pa = pr < 3 ? 0 : (pr < 5 ? 1 : 2);
  const int pb = pr < 3 ? pr + 1 : (pr < 5 ? pr - 1 : 3);
  bool emit = false;
  unsigned a = 0, b = 0;
  if (t < T) {
    a = (unsigned)tets[t * 4 + pa];
    b = (unsigned)tets[t * 4 + pb];
    emit = comp[a] != comp[b];
  }
  const u64 mask = __ballot(emit);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  u64 base = 0;
  if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(counter, (u64)(2 * __popcll(mask)));
  base = __shfl(base, __ffsll((long long)mask) - 1, 64);
  if (emit && keys != nullptr) {
    const u64 pos = base + 2 * (u64)__popcll(mask & ((1ull << lane) - 1ull));
    if ((long)pos + 1 < capacity) {
      keys[pos] = ((u64)a << 32) | b;
      keys[pos + 1] = ((u64)b << 32) | a;
    }
  }
}

// first of a run of equal keys AND (d_max <= 0 OR ||xyz[a] - xyz[b]|| < d_max), the distance in float32 exactly as
// np.sqrt(((xyz[e0] - xyz[e1]) ** 2).sum(1)) evaluates it: three rounded products, added left to right, correctly rounded root
__global__ void edge_flags_kernel(const u64* __restrict__ keys, long n, const float* __restrict__ xyz, float d_max,
                                  unsigned char* __restrict__ flags) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 k = keys[i];
  bool keep = i == 0 || keys[i - 1] != k;
  if (keep && d_max > 0.f) {
    const long a = (long)(k >> 32), b = (long)(k & 0xffffffffull);
    const float dx = __fsub_rn(xyz[3 * a], xyz[3 * b]), dy = __fsub_rn(xyz[3 * a + 1], xyz[3 * b + 1]),
                dz = __fsub_rn(xyz[3 * a + 2], xyz[3 * b + 2]);
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    keep = __fsqrt_rn(s) < d_max;
  }
  flags[i] = keep ? 1 : 0;
}

__global__ void cc_keys_kernel(const u64* __restrict__ edge_keys, const u64* __restrict__ count, const int32_t* __restrict__ comp,
                               long n_com, u64* __restrict__ cc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)*count) return;
  const u64 k = edge_keys[i];
  cc[i] = (u64)comp[k >> 32] * (u64)n_com + (u64)comp[k & 0xffffffffull];      // graphs.py:120
}

__global__ void copy_count_kernel(const unsigned* __restrict__ in, int64_t* __restrict__ out) { *out = (int64_t)*in; }

// -------------------------------------------------------------------------------------------------------------------
// superpoints
// -------------------------------------------------------------------------------------------------------------------
__global__ void point_keys_yz_kernel(const float* __restrict__ xyz, long n, u64* __restrict__ keys, unsigned* __restrict__ idx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = ((u64)ordered_bits(xyz[3 * i + 1]) << 32) | ordered_bits(xyz[3 * i + 2]);
  idx[i] = (unsigned)i;
}

__global__ void point_keys_cx_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ comp, const unsigned* __restrict__ idx,
                                     long n, u64* __restrict__ keys) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned i = idx[j];
  keys[j] = ((u64)(unsigned)comp[i] << 32) | ordered_bits(xyz[3 * (long)i]);
}

// off[c] = first position whose component is >= c (c = 0 .. n_com): lower bound in the sorted keys
__global__ void comp_offsets_kernel(const u64* __restrict__ keys, long n, int n_com, int64_t* __restrict__ off) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_com) return;
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if ((long)(keys[mid] >> 32) < (long)c) lo = mid + 1; else hi = mid;
  }
  off[c] = lo;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);      // xor butterfly: every lane gets the same fixed-order sum
  return v;
}

__device__ __forceinline__ bool same_point(const float* __restrict__ xyz, unsigned a, unsigned b) {
  return xyz[3 * (long)a] == xyz[3 * (long)b] && xyz[3 * (long)a + 1] == xyz[3 * (long)b + 1] &&
         xyz[3 * (long)a + 2] == xyz[3 * (long)b + 2];
}

// eigenvalues of a symmetric 3x3 matrix (a00 a01 a02 a11 a12 a22) by cyclic Jacobi rotations, float64, descending
__device__ void jacobi_eigenvalues(double a00, double a01, double a02, double a11, double a12, double a22, double ev[3]) {
  double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double offd = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (offd == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const int r = 3 - p - q;
        const double app = A[p][p], aqq = A[q][q], apq = A[p][q], arp = A[r][p], arq = A[r][q];
        A[p][p] = app - t * apq;
        A[q][q] = aqq + t * apq;
        A[p][q] = A[q][p] = 0.0;
        A[r][p] = A[p][r] = c * arp - s * arq;
        A[r][q] = A[q][r] = s * arp + c * arq;
      }
  }
  double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2], tmp;
  if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
  if (e1 < e2) { tmp = e1; e1 = e2; e2 = tmp; }
  if (e0 < e1) { tmp = e0; e0 = e1; e1 = tmp; }
  ev[0] = e0; ev[1] = e1; ev[2] = e2;
}

// one wavefront per component (graphs.py:141-172).  idx: point indices ordered by (component, x, y, z); off: component segments
__global__ __launch_bounds__(256) void superpoints_kernel(const float* __restrict__ xyz, const unsigned* __restrict__ idx,
                                                          const int64_t* __restrict__ off, int n_com, float* __restrict__ centroids,
                                                          float* __restrict__ length, float* __restrict__ surface,
                                                          float* __restrict__ volume, u64* __restrict__ point_count) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n_com) return;
  const long b = off[c], e = off[c + 1];
  // pass 1: unique points (first of every run of equal rows) and their sum
  double m = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
  for (long j = b + lane; j < e; j += 64) {
    const unsigned p = idx[j];
    if (j == b || !same_point(xyz, p, idx[j - 1])) {
      m += 1.0; sx += (double)xyz[3 * (long)p]; sy += (double)xyz[3 * (long)p + 1]; sz += (double)xyz[3 * (long)p + 2];
    }
  }
  m = wave_sum_d(m); sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
  if (lane == 0) point_count[c] = (u64)(e - b);      // graphs.py:149 (duplicates included)
  if (m < 0.5) {                                      // empty component: the reference's loop would fail; zeros
    if (lane == 0) {
      centroids[3 * c] = centroids[3 * c + 1] = centroids[3 * c + 2] = 0.f;
      length[c] = surface[c] = volume[c] = 0.f;
    }
    return;
  }
  if (m < 1.5) {                                      // graphs.py:151-155
    if (lane == 0) {
      const unsigned p = idx[b];
      centroids[3 * c] = xyz[3 * (long)p]; centroids[3 * c + 1] = xyz[3 * (long)p + 1]; centroids[3 * c + 2] = xyz[3 * (long)p + 2];
      length[c] = surface[c] = volume[c] = 0.f;
    }
    return;
  }
  if (m < 2.5) {                                      // graphs.py:156-160, float32 throughout like np.mean / np.var of two rows
    if (lane == 0) {
      const unsigned p = idx[b];
      long j = b + 1;
      while (same_point(xyz, idx[j], p)) ++j;         // the second unique point
      const unsigned q = idx[j];
      float var = 0.f;
      for (int d = 0; d < 3; ++d) {
        const float u = xyz[3 * (long)p + d], v = xyz[3 * (long)q + d];
        const float mean = __fdiv_rn(__fadd_rn(u, v), 2.f);
        centroids[3 * c + d] = mean;
        const float du = __fsub_rn(u, mean), dv = __fsub_rn(v, mean);
        const float vd = __fdiv_rn(__fadd_rn(__fmul_rn(du, du), __fmul_rn(dv, dv)), 2.f);
        var = d == 0 ? vd : __fadd_rn(var, vd);
      }
      length[c] = __fsqrt_rn(var);
      surface[c] = volume[c] = 0.f;
    }
    return;
  }
  const double mx = sx / m, my = sy / m, mz = sz / m;
  // pass 2: centred second moments of the unique points (np.cov: float64, divisor m - 1)
  double cxx = 0.0, cxy = 0.0, cxz = 0.0, cyy = 0.0, cyz = 0.0, czz = 0.0;
  for (long j = b + lane; j < e; j += 64) {
    const unsigned p = idx[j];
    if (j == b || !same_point(xyz, p, idx[j - 1])) {
      const double dx = (double)xyz[3 * (long)p] - mx, dy = (double)xyz[3 * (long)p + 1] - my, dz = (double)xyz[3 * (long)p + 2] - mz;
      cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
    }
  }
  cxx = wave_sum_d(cxx); cxy = wave_sum_d(cxy); cxz = wave_sum_d(cxz); cyy = wave_sum_d(cyy); cyz = wave_sum_d(cyz); czz = wave_sum_d(czz);
  if (lane == 0) {
    const double f = 1.0 / (m - 1.0);
    double ev[3];
    jacobi_eigenvalues(cxx * f, cxy * f, cxz * f, cyy * f, cyz * f, czz * f, ev);
    centroids[3 * c] = (float)mx; centroids[3 * c + 1] = (float)my; centroids[3 * c + 2] = (float)mz;
    length[c] = (float)ev[0];                                    // graphs.py:165
    surface[c] = (float)sqrt(ev[0] * ev[1] + 1e-10);             // :169
    volume[c] = (float)sqrt(ev[0] * ev[1] * ev[2] + 1e-10);      // :173
  }
}

// sp_labels (graphs.py:144-148): 1-D labels -> histogram over bins [-0.5, 0.5, ..., n_labels + 0.5); 2-D -> column sums
__global__ void label_hist_kernel(const int32_t* __restrict__ comp, long n, const int32_t* __restrict__ labels,
                                  const uint32_t* __restrict__ label_rows, int n_labels, uint32_t* __restrict__ sp_labels) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long row = (long)comp[i] * (n_labels + 1);
  if (labels != nullptr) {
    const int v = labels[i];
    if (v >= 0 && v <= n_labels) atomicAdd(&sp_labels[row + v], 1u);
  } else {
    for (int k = 0; k <= n_labels; ++k) {
      const uint32_t v = label_rows[i * (n_labels + 1) + k];
      if (v) atomicAdd(&sp_labels[row + k], v);
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------
// superedges (graphs.py:174-208), one wavefront each
// -------------------------------------------------------------------------------------------------------------------
struct SuperedgeOut {
  uint32_t *source, *target;
  float *delta_mean, *delta_std, *delta_norm, *delta_centroid, *length_ratio, *surface_ratio, *volume_ratio, *point_count_ratio;
};

__global__ __launch_bounds__(256) void superedges_kernel(const u64* __restrict__ edges, const u64* __restrict__ seg_cc,
                                                         const int64_t* __restrict__ seg_off, long n_sedg, long n_com,
                                                         const float* __restrict__ xyz, const float* __restrict__ centroids,
                                                         const float* __restrict__ length, const float* __restrict__ surface,
                                                         const float* __restrict__ volume, const u64* __restrict__ point_count,
                                                         SuperedgeOut o) {
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= n_sedg) return;
  const long b = seg_off[s], e = seg_off[s + 1];
  const u64 cc = seg_cc[s];
  const long cs = (long)(cc / (u64)n_com), ct = (long)(cc % (u64)n_com);
  double sd[3] = {0.0, 0.0, 0.0}, sn = 0.0;
  for (long j = b + lane; j < e; j += 64) {
    const u64 k = edges[j];
    const long a = (long)(k >> 32), t = (long)(k & 0xffffffffull);
    const float dx = __fsub_rn(xyz[3 * a], xyz[3 * t]), dy = __fsub_rn(xyz[3 * a + 1], xyz[3 * t + 1]),
                dz = __fsub_rn(xyz[3 * a + 2], xyz[3 * t + 2]);      // delta is a float32 array in the reference (:193)
    sd[0] += (double)dx; sd[1] += (double)dy; sd[2] += (double)dz;
    sn += (double)__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  }
  const double cnt = (double)(e - b);
  double mean[3];
  for (int d = 0; d < 3; ++d) mean[d] = wave_sum_d(sd[d]) / cnt;
  sn = wave_sum_d(sn);
  double var[3] = {0.0, 0.0, 0.0};
  if (e - b > 1) {
    for (long j = b + lane; j < e; j += 64) {
      const u64 k = edges[j];
      const long a = (long)(k >> 32), t = (long)(k & 0xffffffffull);
      for (int d = 0; d < 3; ++d) {
        const double dv = (double)__fsub_rn(xyz[3 * a + d], xyz[3 * t + d]) - mean[d];
        var[d] += dv * dv;
      }
    }
    for (int d = 0; d < 3; ++d) var[d] = wave_sum_d(var[d]) / cnt;      // np.std: ddof = 0
  }
  if (lane != 0) return;
  o.source[s] = (uint32_t)cs;
  o.target[s] = (uint32_t)ct;
  for (int d = 0; d < 3; ++d) {
    o.delta_centroid[3 * s + d] = __fsub_rn(centroids[3 * cs + d], centroids[3 * ct + d]);      // :186
    o.delta_mean[3 * s + d] = (float)mean[d];                                                  // :195 / :199 (one edge: the offset itself)
    o.delta_std[3 * s + d] = (e - b > 1) ? (float)sqrt(var[d]) : 0.f;                          // :196 / :200
  }
  o.delta_norm[s] = (float)(sn / cnt);                                                         // :197 / :201
  o.length_ratio[s] = __fdiv_rn(length[cs], __fadd_rn(length[ct], 1e-6f));                     // :187 (float32 arrays, weak python scalar)
  o.surface_ratio[s] = __fdiv_rn(surface[cs], __fadd_rn(surface[ct], 1e-6f));                  // :188
  o.volume_ratio[s] = __fdiv_rn(volume[cs], __fadd_rn(volume[ct], 1e-6f));                     // :189
  o.point_count_ratio[s] = (float)((double)point_count[cs] / ((double)point_count[ct] + 1e-6));   // :190 (uint64 -> float64)
}

// -------------------------------------------------------------------------------------------------------------------
// compute_geof (partition/ply_c/ply_c.cpp:384-462): one thread per point
// -------------------------------------------------------------------------------------------------------------------
__device__ void jacobi_eigen3(double A[3][3], double V[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double offd = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (offd == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        const int r = 3 - p - q;
        const double apq = A[p][q], arp = A[r][p], arq = A[r][q];
        A[p][p] -= t * apq;
        A[q][q] += t * apq;
        A[p][q] = A[q][p] = 0.0;
        A[r][p] = A[p][r] = c * arp - s * arq;
        A[r][q] = A[q][r] = s * arp + c * arq;
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

// 256 points per workgroup; their k_nn neighbour indices are staged through LDS with coalesced loads (a thread's own row is a
// 4 k_nn-byte strided segment of `target`: read directly, every load instruction would touch 64 different cache lines).
// One gather pass: moments of the offsets from the point itself (d = x_j - x_i, exact in float64), covariance = E[d d^T] - E[d] E[d]^T.
__global__ __launch_bounds__(256) void geof_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ target, long n, int k_nn,
                                                   float* __restrict__ geof) {
  extern __shared__ uint32_t nb[];                       // [256][k_nn + 1] (row padded by one word: conflict-free per-thread rows)
  const long base = (long)blockIdx.x * 256;
  const int rows = (int)min(256L, n - base), ld = k_nn + 1;
  for (long u = threadIdx.x; u < (long)rows * k_nn; u += 256) {
    const int r = (int)(u / k_nn), c = (int)(u - (long)r * k_nn);
    nb[r * ld + c] = target[base * k_nn + u];
  }
  __syncthreads();
  const long i = base + threadIdx.x;
  if (i >= n) return;
  // neighbourhood = the point and its k_nn neighbours (:398-412); centred second moments / (k_nn + 1) (:414-415)
  const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  double sx = 0, sy = 0, sz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
  const uint32_t* mine = nb + threadIdx.x * ld;
  for (int k = 0; k < k_nn; ++k) {
    const long j = mine[k];
    const double dx = (double)xyz[3 * j] - px, dy = (double)xyz[3 * j + 1] - py, dz = (double)xyz[3 * j + 2] - pz;
    sx += dx; sy += dy; sz += dz;
    xx += dx * dx; xy += dx * dy; xz += dx * dz; yy += dy * dy; yz += dy * dz; zz += dz * dz;
  }
  const double inv = 1.0 / (double)(k_nn + 1);
  const double mx = sx * inv, my = sy * inv, mz = sz * inv;      // (the point itself contributes d = 0)
  double A[3][3];
  A[0][0] = xx * inv - mx * mx; A[0][1] = xy * inv - mx * my; A[0][2] = xz * inv - mx * mz;
  A[1][1] = yy * inv - my * my; A[1][2] = yz * inv - my * mz; A[2][2] = zz * inv - mz * mz;
  A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
  double V[3][3];
  jacobi_eigen3(A, V);
  int o0 = 0, o1 = 1, o2 = 2, tmp;      // indices by descending eigenvalue (:420-424)
  if (A[o0][o0] < A[o1][o1]) { tmp = o0; o0 = o1; o1 = tmp; }
  if (A[o1][o1] < A[o2][o2]) { tmp = o1; o1 = o2; o2 = tmp; }
  if (A[o0][o0] < A[o1][o1]) { tmp = o0; o0 = o1; o1 = tmp; }
  const double l0 = fmax(A[o0][o0], 0.0), l1 = fmax(A[o1][o1], 0.0), l2 = fmax(A[o2][o2], 0.0);
  const double s0 = sqrt(l0), s1 = sqrt(l1), s2 = sqrt(l2);
  double u[3];
  for (int d = 0; d < 3; ++d) u[d] = l0 * fabs(V[d][o0]) + l1 * fabs(V[d][o1]) + l2 * fabs(V[d][o2]);      // :441-444
  const double norm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  geof[4 * i + 0] = (float)((s0 - s1) / s0);      // linearity  (:437)
  geof[4 * i + 1] = (float)((s1 - s2) / s0);      // planarity  (:438)
  geof[4 * i + 2] = (float)(s2 / s0);             // scattering (:439)
  geof[4 * i + 3] = (float)(u[2] / norm);         // verticality (:447)
}

struct Carve {
  char* p;
  size_t left;
  bool ok = true;
  void* take(size_t bytes) {
    bytes = align256(bytes);
    if (bytes > left) { ok = false; return nullptr; }
    void* r = p;
    p += bytes; left -= bytes;
    return r;
  }
};

size_t sort_keys_tmp(long n) {
  size_t b = 0;
  (void)rocprim::radix_sort_keys(nullptr, b, (u64*)nullptr, (u64*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  return b;
}
size_t sort_pairs_u64_tmp(long n) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  return b;
}
size_t sort_pairs_u32_tmp(long n) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (u64*)nullptr, (u64*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  return b;
}
size_t select_tmp(long n) {
  size_t b = 0;
  (void)rocprim::select(nullptr, b, (u64*)nullptr, (unsigned char*)nullptr, (u64*)nullptr, (u64*)nullptr, (size_t)n, (hipStream_t)0);
  return b;
}
size_t rle_tmp(long n) {
  size_t b = 0;
  (void)rocprim::run_length_encode(nullptr, b, (u64*)nullptr, (unsigned)n, (u64*)nullptr, (int64_t*)nullptr, (unsigned*)nullptr, (hipStream_t)0);
  return b;
}
size_t scan_tmp(long n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), (hipStream_t)0);
  return b;
}
size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// -------------------------------------------------------------------------------------------------------------------
// prune (partition/ply_c/ply_c.cpp:288-382): voxel-grid subsampling -- mean position / mean colour / label and object
// histograms per non-empty voxel, voxels numbered in the order of their FIRST point (the reference's std::map insertion counter).
// Bit-exact by construction: the bin of a point is floor((x - x_min) / voxel_size) in float32 exactly as the reference writes
// it; a stable radix sort by (bin_x, bin_y, bin_z) keeps the points of a voxel in input order, so ONE thread per voxel adds its
// positions in the reference's order with float32 additions (no reassociation, no fma) -- the same rounding as the serial loop.
// -------------------------------------------------------------------------------------------------------------------
__global__ void minmax_kernel(const float* __restrict__ xyz, long n, unsigned* __restrict__ mm) {      // mm[0..2] = min, mm[3..5] = max (ordered bits)
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    for (int d = 0; d < 3; ++d) { const float v = xyz[3 * i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[d], ordered_bits(lo[d])); atomicMax(&mm[3 + d], ordered_bits(hi[d])); }
  }
}
__device__ __forceinline__ float from_ordered(unsigned b) { return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b); }

__global__ void voxel_keys_kernel(const float* __restrict__ xyz, long n, const unsigned* __restrict__ mm, float voxel, u64* __restrict__ keys,
                                  unsigned* __restrict__ idx, unsigned* __restrict__ flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 k = 0;
  for (int d = 0; d < 3; ++d) {
    const float b = floorf(__fdiv_rn(__fsub_rn(xyz[3 * i + d], from_ordered(mm[d])), voxel));      // ply_c.cpp:326-328 / 338-340
    const unsigned u = (unsigned)b;
    if (!(b >= 0.f) || u >= (1u << 21)) atomicOr(flag, 1u);                                          // grid wider than 2^21 bins per axis
    k = (k << 21) | (u64)(u & 0x1fffffu);
  }
  keys[i] = k;
  idx[i] = (unsigned)i;
}

// key of segment v = index of its first point (the stable sort keeps input order inside a voxel: the first entry of a segment is
// the voxel's first occurrence); the slots behind the last segment (their number is only known on the device) get keys >= 2^32
__global__ void segment_first_kernel(const unsigned* __restrict__ idx_sorted, const int64_t* __restrict__ seg_off, const unsigned* __restrict__ nseg,
                                     long n, u64* __restrict__ first, unsigned* __restrict__ seg_id, int64_t* __restrict__ n_voxels,
                                     const unsigned* __restrict__ flag, int32_t* __restrict__ error_flag) {
  const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v == 0) { *n_voxels = (int64_t)*nseg; if (error_flag != nullptr) *error_flag = (int32_t)*flag; }
  if (v >= n) return;
  first[v] = v < (long)*nseg ? (u64)idx_sorted[seg_off[v]] : ((1ull << 32) | (u64)v);
  seg_id[v] = (unsigned)v;
}

struct PruneOut { float* xyz; uint8_t* rgb; uint32_t* labels; uint32_t* objects; int n_labels, n_objects; };

__global__ void prune_reduce_kernel(const float* __restrict__ xyz, const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ labels,
                                    const uint32_t* __restrict__ objects, const unsigned* __restrict__ idx_sorted,
                                    const int64_t* __restrict__ seg_off, const unsigned* __restrict__ order, long n_vox, PruneOut o,
                                    unsigned* __restrict__ flag) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_vox) return;
  const unsigned seg = order[r];
  const long b = seg_off[seg], e = seg_off[seg + 1];
  float ax = 0.f, ay = 0.f, az = 0.f;
  unsigned cr = 0, cg = 0, cb = 0;
  uint32_t* hl = o.labels + r * (o.n_labels + 1);
  uint32_t* ho = o.objects + r * (o.n_objects + 1);
  for (int k = 0; k <= o.n_labels; ++k) hl[k] = 0;
  for (int k = 0; k <= o.n_objects; ++k) ho[k] = 0;
  for (long j = b; j < e; ++j) {                                    // input order (ply_c.cpp:336-354)
    const long p = idx_sorted[j];
    ax = __fadd_rn(ax, xyz[3 * p]); ay = __fadd_rn(ay, xyz[3 * p + 1]); az = __fadd_rn(az, xyz[3 * p + 2]);
    if (rgb != nullptr) { cr += rgb[3 * p]; cg += rgb[3 * p + 1]; cb += rgb[3 * p + 2]; }
    if (labels != nullptr) {
      const int l = labels[p];
      if (l <= o.n_labels) hl[l] += 1; else atomicOr(flag, 2u);     // the reference's vector::at would throw
    }
    if (objects != nullptr) {
      const uint32_t q = objects[p];
      if (q <= (uint32_t)o.n_objects) ho[q] += 1; else atomicOr(flag, 2u);
    }
  }
  const float count = (float)(unsigned)(e - b);
  o.xyz[3 * r] = __fdiv_rn(ax, count); o.xyz[3 * r + 1] = __fdiv_rn(ay, count); o.xyz[3 * r + 2] = __fdiv_rn(az, count);      // :365-368
  o.rgb[3 * r] = (uint8_t)__fdiv_rn((float)cr, count); o.rgb[3 * r + 1] = (uint8_t)__fdiv_rn((float)cg, count);               // :371-374
  o.rgb[3 * r + 2] = (uint8_t)__fdiv_rn((float)cb, count);
}

size_t prune_rle_tmp(long n) {
  size_t b = 0;
  (void)rocprim::run_length_encode(nullptr, b, (u64*)nullptr, (unsigned)n, (u64*)nullptr, (int64_t*)nullptr, (unsigned*)nullptr, (hipStream_t)0);
  return b;
}

struct PruneWs {
  unsigned *mm, *nseg, *flag, *idx0, *idx_sorted, *seg_id, *order;
  u64 *keys0, *keys1, *uniq, *first0, *first1;
  int64_t *counts, *seg_off;
  void* tmp; size_t tmp_bytes;
  bool ok;
};
PruneWs prune_carve(void* ws, size_t bytes, long n) {
  Carve w{(char*)ws, bytes};
  PruneWs p;
  p.mm = (unsigned*)w.take(256); p.nseg = p.mm + 8; p.flag = p.mm + 9;
  p.keys0 = (u64*)w.take((size_t)n * 8); p.keys1 = (u64*)w.take((size_t)n * 8);
  p.idx0 = (unsigned*)w.take((size_t)n * 4); p.idx_sorted = (unsigned*)w.take((size_t)n * 4);
  p.uniq = (u64*)w.take((size_t)n * 8);
  p.counts = (int64_t*)w.take((size_t)(n + 1) * 8); p.seg_off = (int64_t*)w.take((size_t)(n + 1) * 8);
  p.first0 = (u64*)w.take((size_t)n * 8); p.first1 = (u64*)w.take((size_t)n * 8);
  p.seg_id = (unsigned*)w.take((size_t)n * 4); p.order = (unsigned*)w.take((size_t)n * 4);
  p.tmp_bytes = max2(max2(sort_pairs_u32_tmp(n), prune_rle_tmp(n)), scan_tmp(n + 1));
  p.tmp = w.take(p.tmp_bytes);
  p.ok = w.ok;
  return p;
}

}  // namespace

extern "C" int spg_spg_tet_edges(const int32_t* tets, long T, const int32_t* comp, uint64_t* keys, long capacity, uint64_t* count,
                                 void* stream) {
  SPG_CHECK_ARG(T >= 0 && count != nullptr && (T == 0 || (tets && comp)) && (keys == nullptr || capacity >= 0), "bad argument");
  hipStream_t st = (hipStream_t)stream;
  SPG_RP(hipMemsetAsync(count, 0, sizeof(uint64_t), st));
  if (T == 0) return 0;
  hipLaunchKernelGGL(tet_edges_kernel, dim3(spg_cdiv(6 * T, 256)), dim3(256), 0, st, tets, T, comp, (u64*)keys, capacity, (u64*)count);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spg_spg_workspace_bytes(int which, long n) {
  if (n < 1) n = 1;
  switch (which) {
    case 0:   // unique_edges: sorted keys + flags + rocPRIM scratch
      return align256((size_t)n * 8) + align256((size_t)n) + align256(max2(sort_keys_tmp(n), select_tmp(n))) + 256;
    case 1:   // group_edges: run lengths + rocPRIM scratch
      return align256((size_t)(n + 1) * 8) + align256(max2(max2(sort_pairs_u64_tmp(n), rle_tmp(n)), scan_tmp(n + 1))) + 512;
    case 2:   // superpoints: two key arrays, two index arrays, component segments (n_com <= n), rocPRIM scratch
      return 2 * align256((size_t)n * 8) + 2 * align256((size_t)n * 4) + align256((size_t)(n + 1) * 8) + align256(sort_pairs_u32_tmp(n)) + 256;
    default:
      return 0;
  }
}

extern "C" int spg_spg_unique_edges(const uint64_t* keys, long n, const float* xyz, const int32_t* comp, long n_com, float d_max,
                                    uint64_t* edge_keys, uint64_t* cc_keys, uint64_t* count, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  SPG_CHECK_ARG(n >= 0 && count && (n == 0 || (keys && xyz && comp && edge_keys && cc_keys && workspace)) && n_com > 0, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { SPG_RP(hipMemsetAsync(count, 0, sizeof(uint64_t), st)); return 0; }
  Carve w{(char*)workspace, workspace_bytes};
  u64* sorted = (u64*)w.take((size_t)n * 8);
  unsigned char* flags = (unsigned char*)w.take((size_t)n);
  size_t tb = max2(sort_keys_tmp(n), select_tmp(n));
  void* tmp = w.take(tb);
  SPG_CHECK_ARG(w.ok, "workspace too small (spg_spg_workspace_bytes(0, n))");
  size_t b = tb;
  SPG_RP(rocprim::radix_sort_keys(tmp, b, (const u64*)keys, sorted, (size_t)n, 0, 64, st));
  hipLaunchKernelGGL(edge_flags_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, st, sorted, n, xyz, d_max, flags);
  SPG_LAUNCH_CHECK();
  b = tb;
  SPG_RP(rocprim::select(tmp, b, sorted, flags, (u64*)edge_keys, (u64*)count, (size_t)n, st));
  hipLaunchKernelGGL(cc_keys_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, st, (const u64*)edge_keys, (const u64*)count, comp, n_com,
                     (u64*)cc_keys);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_spg_group_edges(const uint64_t* cc_keys, const uint64_t* edge_keys, long n, uint64_t* cc_sorted, uint64_t* edges_sorted,
                                   uint64_t* seg_cc, int64_t* seg_off, int64_t* n_seg, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  SPG_CHECK_ARG(n >= 0 && n < (1L << 32) && n_seg && seg_off && (n == 0 || (cc_keys && edge_keys && cc_sorted && edges_sorted && seg_cc && workspace)),
                "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    SPG_RP(hipMemsetAsync(n_seg, 0, sizeof(int64_t), st));
    SPG_RP(hipMemsetAsync(seg_off, 0, sizeof(int64_t), st));
    return 0;
  }
  Carve w{(char*)workspace, workspace_bytes};
  int64_t* counts = (int64_t*)w.take((size_t)(n + 1) * 8);
  unsigned* nruns = (unsigned*)w.take(256);
  size_t tb = max2(max2(sort_pairs_u64_tmp(n), rle_tmp(n)), scan_tmp(n + 1));
  void* tmp = w.take(tb);
  SPG_CHECK_ARG(w.ok, "workspace too small (spg_spg_workspace_bytes(1, n))");
  size_t b = tb;
  SPG_RP(rocprim::radix_sort_pairs(tmp, b, (const u64*)cc_keys, (u64*)cc_sorted, (const u64*)edge_keys, (u64*)edges_sorted, (size_t)n, 0, 64, st));
  SPG_RP(hipMemsetAsync(counts, 0, (size_t)(n + 1) * 8, st));
  b = tb;
  SPG_RP(rocprim::run_length_encode(tmp, b, (const u64*)cc_sorted, (unsigned)n, (u64*)seg_cc, counts, nruns, st));
  b = tb;
  SPG_RP(rocprim::exclusive_scan(tmp, b, counts, seg_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(copy_count_kernel, dim3(1), dim3(1), 0, st, (const unsigned*)nruns, n_seg);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_spg_superpoints(const float* xyz, long n, const int32_t* comp, int n_com, const int32_t* labels,
                                   const uint32_t* label_rows, int n_labels, float* centroids, float* length, float* surface,
                                   float* volume, uint64_t* point_count, uint32_t* sp_labels, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  SPG_CHECK_ARG(n > 0 && n < (1L << 32) && n_com > 0 && n_com <= n && xyz && comp && centroids && length && surface && volume && point_count && workspace,
                "bad argument");
  SPG_CHECK_ARG(!(labels && label_rows) && ((labels == nullptr && label_rows == nullptr) || (sp_labels && n_labels >= 0)), "bad label arguments");
  hipStream_t st = (hipStream_t)stream;
  Carve w{(char*)workspace, workspace_bytes};
  u64* k0 = (u64*)w.take((size_t)n * 8);
  u64* k1 = (u64*)w.take((size_t)n * 8);
  unsigned* i0 = (unsigned*)w.take((size_t)n * 4);
  unsigned* i1 = (unsigned*)w.take((size_t)n * 4);
  int64_t* offs = (int64_t*)w.take((size_t)(n_com + 1) * 8);      // component segments of the ordered points
  size_t tb = sort_pairs_u32_tmp(n);
  void* tmp = w.take(tb);
  SPG_CHECK_ARG(w.ok, "workspace too small (spg_spg_workspace_bytes(2, n))");
  const dim3 grid(spg_cdiv(n, 256)), block(256);
  hipLaunchKernelGGL(point_keys_yz_kernel, grid, block, 0, st, xyz, n, k0, i0);
  SPG_LAUNCH_CHECK();
  size_t b = tb;
  SPG_RP(rocprim::radix_sort_pairs(tmp, b, (const u64*)k0, k1, (const unsigned*)i0, i1, (size_t)n, 0, 64, st));      // by (y, z)
  hipLaunchKernelGGL(point_keys_cx_kernel, grid, block, 0, st, xyz, comp, (const unsigned*)i1, n, k0);
  SPG_LAUNCH_CHECK();
  b = tb;
  SPG_RP(rocprim::radix_sort_pairs(tmp, b, (const u64*)k0, k1, (const unsigned*)i1, i0, (size_t)n, 0, 64, st));      // stable: (comp, x, y, z)
  hipLaunchKernelGGL(comp_offsets_kernel, dim3(spg_cdiv(n_com + 1, 256)), block, 0, st, (const u64*)k1, n, n_com, offs);
  SPG_LAUNCH_CHECK();
  hipLaunchKernelGGL(superpoints_kernel, dim3(spg_cdiv(n_com, 4)), block, 0, st, xyz, (const unsigned*)i0, (const int64_t*)offs, n_com,
                     centroids, length, surface, volume, (u64*)point_count);
  SPG_LAUNCH_CHECK();
  if (labels != nullptr || label_rows != nullptr) {
    SPG_RP(hipMemsetAsync(sp_labels, 0, (size_t)n_com * (n_labels + 1) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(label_hist_kernel, grid, block, 0, st, comp, n, labels, label_rows, n_labels, sp_labels);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int spg_spg_superedges(const uint64_t* edges_sorted, const uint64_t* seg_cc, const int64_t* seg_off, long n_sedg, long n_com,
                                  const float* xyz, const float* centroids, const float* length, const float* surface, const float* volume,
                                  const uint64_t* point_count, uint32_t* source, uint32_t* target, float* delta_mean, float* delta_std,
                                  float* delta_norm, float* delta_centroid, float* length_ratio, float* surface_ratio, float* volume_ratio,
                                  float* point_count_ratio, void* stream) {
  SPG_CHECK_ARG(n_sedg >= 0 && n_com > 0, "bad argument");
  if (n_sedg == 0) return 0;
  SPG_CHECK_ARG(edges_sorted && seg_cc && seg_off && xyz && centroids && length && surface && volume && point_count && source && target &&
                    delta_mean && delta_std && delta_norm && delta_centroid && length_ratio && surface_ratio && volume_ratio && point_count_ratio,
                "null argument");
  SuperedgeOut o{source, target, delta_mean, delta_std, delta_norm, delta_centroid, length_ratio, surface_ratio, volume_ratio, point_count_ratio};
  hipLaunchKernelGGL(superedges_kernel, dim3(spg_cdiv(n_sedg, 4)), dim3(256), 0, (hipStream_t)stream, (const u64*)edges_sorted,
                     (const u64*)seg_cc, seg_off, n_sedg, n_com, xyz, centroids, length, surface, volume, (const u64*)point_count, o);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_compute_geof(const float* xyz, const uint32_t* target, long n, int k_nn, float* geof, void* stream) {
  SPG_CHECK_ARG(n >= 0 && k_nn >= 0 && (n == 0 || (xyz && geof && (k_nn == 0 || target))), "bad argument");
  if (n == 0) return 0;
  SPG_CHECK_ARG(k_nn <= 150, "k_nn too large for the LDS staging of the neighbour lists (150)");
  hipLaunchKernelGGL(geof_kernel, dim3(spg_cdiv(n, 256)), dim3(256), (size_t)256 * (k_nn + 1) * sizeof(uint32_t), (hipStream_t)stream, xyz,
                     target, n, k_nn, geof);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spg_prune_workspace_bytes(long n) {
  if (n < 1) n = 1;
  return 256 + 5 * align256((size_t)n * 8) + 4 * align256((size_t)n * 4) + 2 * align256((size_t)(n + 1) * 8) +
         align256(max2(max2(sort_pairs_u32_tmp(n), prune_rle_tmp(n)), scan_tmp(n + 1))) + 4096;
}

// phase 1: voxel of every point, points ordered by voxel, voxels in first-occurrence order; *n_voxels (device int64) for the caller
extern "C" int spg_prune_voxels(const float* xyz, long n, float voxel_size, int64_t* n_voxels, int32_t* error_flag, void* workspace,
                                size_t workspace_bytes, void* stream) {
  SPG_CHECK_ARG(xyz && n > 0 && n < (1L << 32) && voxel_size > 0.f && n_voxels && workspace, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  PruneWs p = prune_carve(workspace, workspace_bytes, n);
  SPG_CHECK_ARG(p.ok, "workspace too small (spg_prune_workspace_bytes(n))");
  SPG_RP(hipMemsetAsync(p.mm, 0xff, 3 * sizeof(unsigned), st));          // running minima (ordered bits): all ones
  SPG_RP(hipMemsetAsync(p.mm + 3, 0, 7 * sizeof(unsigned), st));          // running maxima, segment count, flag: zero
  const dim3 block(256);
  hipLaunchKernelGGL(minmax_kernel, dim3(n < 262144 ? spg_cdiv(n, 256) : 1024), block, 0, st, xyz, n, p.mm);
  SPG_LAUNCH_CHECK();
  hipLaunchKernelGGL(voxel_keys_kernel, dim3(spg_cdiv(n, 256)), block, 0, st, xyz, n, (const unsigned*)p.mm, voxel_size, p.keys0, p.idx0, p.flag);
  SPG_LAUNCH_CHECK();
  size_t b = p.tmp_bytes;
  SPG_RP(rocprim::radix_sort_pairs(p.tmp, b, (const u64*)p.keys0, p.keys1, (const unsigned*)p.idx0, p.idx_sorted, (size_t)n, 0, 63, st));
  SPG_RP(hipMemsetAsync(p.counts, 0, (size_t)(n + 1) * 8, st));
  b = p.tmp_bytes;
  SPG_RP(rocprim::run_length_encode(p.tmp, b, (const u64*)p.keys1, (unsigned)n, p.uniq, p.counts, p.nseg, st));
  b = p.tmp_bytes;
  SPG_RP(rocprim::exclusive_scan(p.tmp, b, p.counts, p.seg_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), st));
  // voxels in the order of their first point: sort the segments by the index of their first entry
  hipLaunchKernelGGL(segment_first_kernel, dim3(spg_cdiv(n, 256)), block, 0, st, (const unsigned*)p.idx_sorted, (const int64_t*)p.seg_off,
                     (const unsigned*)p.nseg, n, p.first0, p.seg_id, n_voxels, (const unsigned*)p.flag, error_flag);
  SPG_LAUNCH_CHECK();
  b = p.tmp_bytes;
  SPG_RP(rocprim::radix_sort_pairs(p.tmp, b, (const u64*)p.first0, p.first1, (const unsigned*)p.seg_id, p.order, (size_t)n, 0, 34, st));
  return 0;
}

// phase 2 (same workspace, untouched since phase 1): per-voxel means and histograms into arrays of n_voxels rows.
// rgb uint8 [n,3] or NULL (zeros), labels uint8 [n] or NULL, objects uint32 [n] or NULL; error_flag |= 2 for a label / object id
// beyond n_labels / n_objects (the reference's vector::at would throw).
extern "C" int spg_prune_reduce(const float* xyz, const uint8_t* rgb, const uint8_t* labels, const uint32_t* objects, long n, long n_voxels,
                                int n_labels, int n_objects, float* out_xyz, uint8_t* out_rgb, uint32_t* out_labels, uint32_t* out_objects,
                                int32_t* error_flag, void* workspace, size_t workspace_bytes, void* stream) {
  SPG_CHECK_ARG(xyz && n > 0 && n_voxels >= 0 && n_voxels <= n && n_labels >= 0 && n_objects >= 0 && workspace, "bad argument");
  if (n_voxels == 0) return 0;
  SPG_CHECK_ARG(out_xyz && out_rgb && out_labels && out_objects, "null output");
  hipStream_t st = (hipStream_t)stream;
  PruneWs p = prune_carve(workspace, workspace_bytes, n);
  SPG_CHECK_ARG(p.ok, "workspace too small (spg_prune_workspace_bytes(n))");
  PruneOut o{out_xyz, out_rgb, out_labels, out_objects, n_labels, n_objects};
  hipLaunchKernelGGL(prune_reduce_kernel, dim3(spg_cdiv(n_voxels, 256)), dim3(256), 0, st, xyz, rgb, labels, objects,
                     (const unsigned*)p.idx_sorted, (const int64_t*)p.seg_off, (const unsigned*)p.order, n_voxels, o, p.flag);
  SPG_LAUNCH_CHECK();
  if (error_flag != nullptr) {
    SPG_RP(hipMemcpyAsync(error_flag, p.flag, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  }
  return 0;
}
