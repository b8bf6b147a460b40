// C-ABI entry points of libspg_hip.so that are thin wrappers over the kernels (graph build, generic ECC
// operator for arbitrary channel counts / fp64, stand-alone GRU cell, fused dense layer).
// The network-level entry points live in spg_pointnet.hip and spg_eccnet.hip.
#include "../../include/spg_hip.h"
#include "spg_ecc.h"
#include "spg_gemm.h"
#include <mutex>

extern "C" int spg_version(void) { return SPG_VERSION; }

// ---------------------------------------------------------------------------------------------
// side stream (spg_common.h)
// ---------------------------------------------------------------------------------------------
namespace {
struct SpgSide { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false, tried = false; };
SpgSide g_side[SPG_MAX_DEVICES];
std::mutex g_side_mutex;
SpgSide* side_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SPG_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_side_mutex);
  SpgSide& s = g_side[dev];
  if (!s.tried) {
    s.tried = true;
    s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
  }
  return s.ok ? &s : nullptr;
}
}  // namespace

hipStream_t spg_side_fork(hipStream_t main) {
  if (!spg_tune_get(SPG_TUNE_SIDE_STREAM)) return nullptr;      // opt-in: measured slower (spg_common.h)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;   // hipGraph capture: stay on one stream
  SpgSide* s = side_of_current_device();
  if (s == nullptr) return nullptr;
  if (hipEventRecord(s->fork, main) != hipSuccess || hipStreamWaitEvent(s->stream, s->fork, 0) != hipSuccess) return nullptr;
  return s->stream;
}

int spg_side_join(hipStream_t main) {
  SpgSide* s = side_of_current_device();
  if (s == nullptr) return 0;
  hipError_t e = hipEventRecord(s->join, s->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(main, s->join, 0);
  if (e != hipSuccess) { spg_set_error("side stream join: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------------
extern "C" size_t spg_graph_workspace_bytes(int N, int n_src, int E) { return spg_graph_bytes(N, n_src, E); }

extern "C" int spg_graph_build(const int64_t* idxn, const int64_t* degs, int N, int n_src, int E, void* graph_ws,
                               void* stream) {
  SPG_CHECK_ARG(degs && graph_ws && (E == 0 || idxn), "null pointer");
  return spg_graph_build_impl(idxn, degs, N, n_src, E, graph_ws, (hipStream_t)stream);
}

extern "C" int spg_graph_export(const void* graph_ws, int N, int n_src, int E, int32_t* rowptr, int32_t* src, int32_t* dst,
                                int32_t* rev_rowptr, int32_t* rev_eid, int32_t* hdr, void* stream) {
  SPG_CHECK_ARG(graph_ws != nullptr, "null pointer");
  SpgGraph g = spg_graph_view(graph_ws, N, E);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (rowptr) e = hipMemcpyAsync(rowptr, g.rowptr, (size_t)(N + 1) * 4, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && src && E) e = hipMemcpyAsync(src, g.src, (size_t)E * 4, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && dst && E) e = hipMemcpyAsync(dst, g.dst, (size_t)E * 4, hipMemcpyDeviceToDevice, st);
  if (n_src < N) n_src = N;
  if (e == hipSuccess && rev_rowptr) e = hipMemcpyAsync(rev_rowptr, g.rev_rowptr, (size_t)(n_src + 1) * 4, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && hdr) e = hipMemcpyAsync(hdr, g.hdr, 16, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && rev_eid && E) e = hipMemcpyAsync(rev_eid, g.rev_eid, (size_t)E * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { spg_set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// generic ECC operator (any channel counts, fp32 / fp64, optional idxe) -- API completeness for
// GraphConvFunction.apply; the 32-channel fp32 hot path uses the fused kernels of spg_ecc.hip.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void spg_ecc_generic_fwd_kernel(SpgGraph g, const T* __restrict__ x, const T* __restrict__ w,
                                           const int64_t* __restrict__ idxe, int cin, int cout, int matrix,
                                           T* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)g.N * cout) return;
  const int i = (int)(t / cout), c = (int)(t - (long)i * cout);
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  T acc = 0;
  for (int e = e0; e < e1; ++e) {
    const long we = idxe ? (long)idxe[e] : (long)e;
    const T* xj = x + (long)g.src[e] * cin;
    if (matrix) {
      const T* W = w + we * (long)cin * cout;
      for (int k = 0; k < cin; ++k) acc += xj[k] * W[(long)k * cout + c];
    } else {
      acc += xj[c] * w[we * cout + c];
    }
  }
  out[t] = e1 > e0 ? acc / (T)(e1 - e0) : (T)0;
}

template <typename T>
__global__ void spg_ecc_generic_bwd_x_kernel(SpgGraph g, const T* __restrict__ w, const int64_t* __restrict__ idxe,
                                             const T* __restrict__ go, int n_x_rows, int cin, int cout, int matrix,
                                             T* __restrict__ gx) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_x_rows * cin) return;
  const int j = (int)(t / cin), k = (int)(t - (long)j * cin);
  T acc = 0;
  if (j < g.hdr[1]) {   // hdr[1] = number of source rows the reverse CSR was built for
    for (int q = g.rev_rowptr[j]; q < g.rev_rowptr[j + 1]; ++q) {
      const int e = g.rev_eid[q];
      const int d = g.dst[e];
      const T inv = (T)1 / (T)(g.rowptr[d + 1] - g.rowptr[d]);
      const long we = idxe ? (long)idxe[e] : (long)e;
      if (matrix) {
        const T* W = w + we * (long)cin * cout + (long)k * cout;
        T s = 0;
        for (int c = 0; c < cout; ++c) s += go[(long)d * cout + c] * W[c];
        acc += s * inv;
      } else {
        acc += go[(long)d * cout + k] * inv * w[we * cout + k];
      }
    }
  }
  gx[t] = acc;
}

template <typename T>
__global__ void spg_ecc_generic_bwd_w_kernel(SpgGraph g, const T* __restrict__ x, const int64_t* __restrict__ idxe,
                                             const T* __restrict__ go, int cin, int cout, int matrix,
                                             T* __restrict__ gw) {
  const long per = matrix ? (long)cin * cout : (long)cout;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)g.E * per) return;
  const int e = (int)(t / per);
  const long rem = t - (long)e * per;
  const int d = g.dst[e];
  const T inv = (T)1 / (T)(g.rowptr[d + 1] - g.rowptr[d]);
  const T* xj = x + (long)g.src[e] * cin;
  T v;
  if (matrix) {
    const int k = (int)(rem / cout), c = (int)(rem - (long)k * cout);
    v = xj[k] * (go[(long)d * cout + c] * inv);
  } else {
    v = xj[rem] * (go[(long)d * cout + rem] * inv);
  }
  if (idxe) atomicAdd(&gw[(long)idxe[e] * per + rem], v);   // filter sharing (never used by the SPG path)
  else gw[t] = v;
}

// ---- stand-alone backward at the hot-path shape (32 -> 32 channels, fp32, no filter sharing): the fused forms of the two
//      generic kernels above.  grad_x: one wavefront per SOURCE node walks its out-edges in the reverse CSR, 4 edges per
//      batch with all loads issued first (a 4 KB filter = 4 float4 per lane), W_e . (grad_out[dst] / deg[dst]) reduced over
//      the 8 lanes that share an input channel -- no atomics, deterministic.  grad_w: one wavefront per edge writes the
//      outer product h_src (x) grad_out[dst] / deg[dst] as 4 float4 per lane.  (reference: GraphConvFunction.backward,
//      learning/ecc/GraphConvModule.py:92-141, which loops over edge shards and calls its CUDA-string kernels.)
__global__ __launch_bounds__(256) void spg_ecc32_bwd_x_kernel(SpgGraph g, const float* __restrict__ w, const float* __restrict__ go,
                                                              int n_x_rows, int matrix, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n_x_rows) return;
  const int b = j < g.hdr[1] ? g.rev_rowptr[j] : 0, e_ = j < g.hdr[1] ? g.rev_rowptr[j + 1] : 0;
  if (matrix) {
    float pq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = b; t < e_; t += 4) {
      int eid[4], did[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) eid[u] = g.rev_eid[min(t + u, e_ - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) did[u] = g.dst[eid[u]];
      f32x4 wv[4][4], g4[4];
      float inv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g4[u] = *reinterpret_cast<const f32x4*>(go + (long)did[u] * 32 + 4 * (lane & 7));
        inv[u] = (t + u < e_) ? 1.f / (float)(g.rowptr[did[u] + 1] - g.rowptr[did[u]]) : 0.f;
        const f32x4* We = reinterpret_cast<const f32x4*>(w + (long)eid[u] * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[u][q] = We[lane + 64 * q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          pq[q] += inv[u] * ((wv[u][q][0] * g4[u][0] + wv[u][q][1] * g4[u][1]) + (wv[u][q][2] * g4[u][2] + wv[u][q][3] * g4[u][3]));
      }
    }
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pq[q] += __shfl_xor(pq[q], off, 64);
    }
    if ((lane & 7) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) gx[(long)j * 32 + (lane >> 3) + 8 * q] = pq[q];      // input channel k = lane/8 + 8q
    }
  } else if (lane < 32) {
    float sacc = 0.f;
    for (int t = b; t < e_; ++t) {
      const int e = g.rev_eid[t], d = g.dst[e];
      sacc = fmaf(w[(long)e * 32 + lane], go[(long)d * 32 + lane] / (float)(g.rowptr[d + 1] - g.rowptr[d]), sacc);
    }
    gx[(long)j * 32 + lane] = sacc;
  }
}

__global__ __launch_bounds__(256) void spg_ecc32_bwd_w_kernel(SpgGraph g, const float* __restrict__ x, const float* __restrict__ go,
                                                              int matrix, float* __restrict__ gw) {
  const int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= g.E) return;
  const int d = g.dst[e];
  const float inv = 1.f / (float)(g.rowptr[d + 1] - g.rowptr[d]);
  const float* hs = x + (long)g.src[e] * 32;
  const float* gd = go + (long)d * 32;
  if (matrix) {
    f32x4 g4 = *reinterpret_cast<const f32x4*>(gd + 4 * (lane & 7));
#pragma unroll
    for (int c = 0; c < 4; ++c) g4[c] *= inv;
    f32x4* o = reinterpret_cast<f32x4*>(gw + (long)e * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float hk = hs[(lane >> 3) + 8 * q];
      o[lane + 64 * q] = f32x4{hk * g4[0], hk * g4[1], hk * g4[2], hk * g4[3]};
    }
  } else if (lane < 32) {
    gw[(long)e * 32 + lane] = hs[lane] * (gd[lane] * inv);
  }
}

static int ecc32_bwd(const float* x, const float* w, const SpgGraph& g, int n_x_rows, int matrix, const float* go, float* gx,
                     float* gw, hipStream_t st) {
  if (gx && n_x_rows > 0) {
    hipLaunchKernelGGL(spg_ecc32_bwd_x_kernel, dim3(spg_cdiv(n_x_rows, 4)), dim3(256), 0, st, g, w, go, n_x_rows, matrix, gx);
    SPG_LAUNCH_CHECK();
  }
  if (gw && g.E > 0) {
    hipLaunchKernelGGL(spg_ecc32_bwd_w_kernel, dim3(spg_cdiv(g.E, 4)), dim3(256), 0, st, g, x, go, matrix, gw);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

template <typename T>
static int ecc_generic_fwd(const void* x, const void* w, const int64_t* idxe, const SpgGraph& g, int cin, int cout,
                           int matrix, void* out, hipStream_t st) {
  const long n = (long)g.N * cout;
  hipLaunchKernelGGL(spg_ecc_generic_fwd_kernel<T>, dim3(spg_cdiv(n, 256)), dim3(256), 0, st, g, (const T*)x, (const T*)w,
                     idxe, cin, cout, matrix, (T*)out);
  SPG_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int ecc_generic_bwd(const void* x, const void* w, const int64_t* idxe, const SpgGraph& g, int n_x_rows,
                           int n_w_rows, int cin, int cout, int matrix, const void* go, void* gx, void* gw,
                           hipStream_t st) {
  if (gx) {
    const long n = (long)n_x_rows * cin;
    hipLaunchKernelGGL(spg_ecc_generic_bwd_x_kernel<T>, dim3(spg_cdiv(n, 256)), dim3(256), 0, st, g, (const T*)w, idxe,
                       (const T*)go, n_x_rows, cin, cout, matrix, (T*)gx);
    SPG_LAUNCH_CHECK();
  }
  if (gw) {
    const long per = matrix ? (long)cin * cout : (long)cout;
    if (idxe) {
      hipError_t e = hipMemsetAsync(gw, 0, (size_t)n_w_rows * per * sizeof(T), st);
      if (e != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    }
    const long n = (long)g.E * per;
    if (n > 0) {
      hipLaunchKernelGGL(spg_ecc_generic_bwd_w_kernel<T>, dim3(spg_cdiv(n, 256)), dim3(256), 0, st, g, (const T*)x, idxe,
                         (const T*)go, cin, cout, matrix, (T*)gw);
      SPG_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int spg_ecc_aggregate_fwd(int dtype, const void* x, const void* w, const int64_t* idxe, const void* graph_ws,
                                     int N, int E, int cin, int cout, int w_is_matrix, void* out, void* stream) {
  SPG_CHECK_ARG(x && out && graph_ws && (E == 0 || w), "null pointer");
  SPG_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (f64)");
  SPG_CHECK_ARG(w_is_matrix || cin == cout, "vector filters need cin == cout");
  SpgGraph g = spg_graph_view(graph_ws, N, E);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0 && cin == 32 && cout == 32 && idxe == nullptr) {   // hot-path shape: fused wave-per-node kernel
    SpgEccStepFwd p; memset(&p, 0, sizeof(p));
    p.g = g; p.W = (const float*)w; p.matrix = w_is_matrix; p.hin = (const float*)x; p.ld = 32;
    p.agg_save = (float*)out; p.ldagg = 32; p.do_gru = 0;
    return spg_launch_ecc_step_fwd(p, st);
  }
  return dtype == 0 ? ecc_generic_fwd<float>(x, w, idxe, g, cin, cout, w_is_matrix, out, st)
                    : ecc_generic_fwd<double>(x, w, idxe, g, cin, cout, w_is_matrix, out, st);
}

extern "C" int spg_ecc_aggregate_bwd(int dtype, const void* x, const void* w, const int64_t* idxe, const void* graph_ws,
                                     int N, int E, int n_x_rows, int n_w_rows, int cin, int cout, int w_is_matrix,
                                     const void* grad_out, void* grad_x, void* grad_w, void* stream) {
  SPG_CHECK_ARG(x && grad_out && graph_ws && (E == 0 || w), "null pointer");
  SPG_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (f64)");
  SpgGraph g = spg_graph_view(graph_ws, N, E);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0 && cin == 32 && cout == 32 && idxe == nullptr &&
      ((((uintptr_t)w) | ((uintptr_t)grad_out) | ((uintptr_t)grad_w)) & 15) == 0)      // hot-path shape: fused kernels
    return ecc32_bwd((const float*)x, (const float*)w, g, n_x_rows, w_is_matrix, (const float*)grad_out, (float*)grad_x,
                     (float*)grad_w, st);
  return dtype == 0 ? ecc_generic_bwd<float>(x, w, idxe, g, n_x_rows, n_w_rows, cin, cout, w_is_matrix, grad_out, grad_x,
                                             grad_w, st)
                    : ecc_generic_bwd<double>(x, w, idxe, g, n_x_rows, n_w_rows, cin, cout, w_is_matrix, grad_out,
                                              grad_x, grad_w, st);
}

// ---------------------------------------------------------------------------------------------
// stand-alone GRUCellEx
// ---------------------------------------------------------------------------------------------
// scratch layout (floats): w_ih_t 3072 | w_hh_t 3072 | w_ig_t 1024 | dgi n*96 | dgh n*96 | dui n*96 | duh n*96 |
//                          dpre n*32 | xg n*32 | G n*32 | dhdir n*32 | wgrad work
static size_t gru_work_floats(int n) { return spg_wgrad_workspace_floats(n, 96, 32); }
extern "C" size_t spg_gru_scratch_floats(int n) { return 7168 + (size_t)n * (4 * 96 + 4 * 32) + gru_work_floats(n) + 64; }

static int gru_pack(const float* const* params, int layernorm, int ingate, float* scratch, SpgGruParams& G, hipStream_t st) {
  memset(&G, 0, sizeof(G));
  G.w_ih = params[0]; G.w_hh = params[1]; G.b_ih = params[2]; G.b_hh = params[3]; G.w_ig = params[4]; G.b_ig = params[5];
  SPG_CHECK_ARG(G.w_ih && G.w_hh && G.b_ih && G.b_hh, "missing GRU parameters");
  SPG_CHECK_ARG(!ingate || (G.w_ig && G.b_ig), "missing input-gate parameters");
  G.layernorm = layernorm; G.ingate = ingate;
  SPG_CHECK_ARG((((uintptr_t)G.w_ih | (uintptr_t)G.w_hh | (uintptr_t)G.w_ig) & 15) == 0, "GRU weights must be 16-byte aligned");
  (void)scratch; (void)st;
  return 0;
}

extern "C" int spg_gru_cell_fwd(const float* input, const float* hidden, int n, const float* const* params, int layernorm,
                                int ingate, float* out, float* scratch, void* stream) {
  SPG_CHECK_ARG(input && hidden && params && out && scratch && n > 0, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  SpgEccStepFwd p; memset(&p, 0, sizeof(p));
  SPG_TRY(gru_pack(params, layernorm, ingate, scratch, p.gru, st));
  p.g.N = n; p.agg_in = input; p.ldagg = 32; p.hin = hidden; p.hout = out; p.ld = 32; p.do_gru = 1;
  return spg_launch_ecc_step_fwd(p, st);
}

extern "C" int spg_gru_cell_bwd(const float* input, const float* hidden, const float* grad_out, int n,
                                const float* const* params, int layernorm, int ingate, float* grad_input,
                                float* grad_hidden, float* const* grads, float* scratch, void* stream) {
  SPG_CHECK_ARG(input && hidden && grad_out && params && grad_input && grad_hidden && grads && scratch && n > 0,
                "null pointer");
  hipStream_t st = (hipStream_t)stream;
  SpgEccStepBwd p; memset(&p, 0, sizeof(p));
  SPG_TRY(gru_pack(params, layernorm, ingate, scratch, p.gru, st));
  float* f = scratch + 7168;
  float* dgi = f; f += (size_t)n * 96;
  float* dgh = f; f += (size_t)n * 96;
  float* dui = f; f += (size_t)n * 96;
  float* duh = f; f += (size_t)n * 96;
  float* dpre = f; f += (size_t)n * 32;
  float* xg = f; f += (size_t)n * 32;
  f += (size_t)n * 64;   // (G, dhdir written straight to the outputs)
  float* work = f;
  p.g.N = n;   // invdeg == nullptr: no degree scaling
  p.dcat = grad_out; p.ldc = 32; p.dhdir = grad_hidden; p.use_dhdir = 0;
  p.hin = hidden; p.ld = 32; p.agg = input; p.ldagg = 32; p.Gcur = grad_input; p.ldg = 32;
  p.dgi = dgi; p.dgh = dgh; p.dui = dui; p.duh = duh; p.ld96 = 96; p.dpre = dpre; p.xg = xg; p.ld32 = 32;
  SPG_TRY(spg_launch_ecc_step_bwd(p, st));
  auto ident = [](const float* X, long ld) { SpgOperand o; memset(&o, 0, sizeof(o)); o.mode = SPG_PRO_IDENT; o.X = X; o.ld = ld; return o; };
  SpgWgradParams w; memset(&w, 0, sizeof(w));
  w.M = n; w.N = 96; w.K = 32;
  if (grads[0]) { w.a = ident(dgi, 96); w.b = ident(xg, 32); SPG_TRY(spg_launch_wgrad(w, grads[0], work, st)); }
  if (grads[1]) { w.a = ident(dgh, 96); w.b = ident(hidden, 32); SPG_TRY(spg_launch_wgrad(w, grads[1], work, st)); }
  if (grads[2]) SPG_TRY(spg_launch_colsum(dui, 96, n, 96, grads[2], work, st));
  if (grads[3]) SPG_TRY(spg_launch_colsum(duh, 96, n, 96, grads[3], work, st));
  if (ingate) {
    w.N = 32;
    if (grads[4]) { w.a = ident(dpre, 32); w.b = ident(hidden, 32); SPG_TRY(spg_launch_wgrad(w, grads[4], work, st)); }
    if (grads[5]) SPG_TRY(spg_launch_colsum(dpre, 32, n, 32, grads[5], work, st));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// stand-alone LSTMCellEx
// ---------------------------------------------------------------------------------------------
// scratch layout (floats): dgi n*128 | dgh n*128 | dpre n*32 | xg n*32 | zero n*32 | wgrad work
extern "C" size_t spg_lstm_scratch_floats(int n) {
  return (size_t)n * (2 * 128 + 3 * 32) + spg_wgrad_workspace_floats(n, 128, 32) + 256;
}

extern "C" int spg_lstm_cell_fwd(const float* input, const float* h, const float* c, int n, const float* const* params,
                                 int layernorm, int ingate, float* hy, float* cy, float* scratch, void* stream) {
  SPG_CHECK_ARG(input && h && params && hy && cy && n > 0, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  SpgEccStepFwd p; memset(&p, 0, sizeof(p));
  SPG_TRY(gru_pack(params, layernorm, ingate, scratch, p.gru, st));
  p.g.N = n; p.agg_in = input; p.ldagg = 32; p.hin = h; p.hout = hy; p.ld = 32; p.do_gru = 1;
  p.cell = SPG_CELL_LSTM; p.cin = c; p.cout = cy;
  return spg_launch_ecc_step_fwd(p, st);
}

extern "C" int spg_lstm_cell_bwd(const float* input, const float* h, const float* c, const float* grad_hy,
                                 const float* grad_cy, int n, const float* const* params, int layernorm, int ingate,
                                 float* grad_input, float* grad_h, float* grad_c, float* const* grads, float* scratch,
                                 void* stream) {
  SPG_CHECK_ARG(input && h && params && grad_input && grad_h && grad_c && grads && scratch && n > 0, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  SpgEccStepBwd p; memset(&p, 0, sizeof(p));
  SPG_TRY(gru_pack(params, layernorm, ingate, scratch, p.gru, st));
  float* f = scratch;
  float* dgi = f; f += (size_t)n * 128;
  float* dgh = f; f += (size_t)n * 128;
  float* dpre = f; f += (size_t)n * 32;
  float* xg = f; f += (size_t)n * 32;
  f += (size_t)n * 32;
  float* work = f;
  p.g.N = n;   // invdeg == nullptr: no degree scaling
  p.dcat = grad_hy; p.ldc = 32; p.dhdir = grad_h; p.use_dhdir = 0;
  p.hin = h; p.ld = 32; p.agg = input; p.ldagg = 32; p.Gcur = grad_input; p.ldg = 32;
  p.dgi = dgi; p.dgh = dgh; p.ld96 = 128; p.dpre = dpre; p.xg = xg; p.ld32 = 32;
  p.cell = SPG_CELL_LSTM; p.cin = c; p.dcdir = grad_c; p.use_dcdir = 0;
  if (grad_cy != nullptr) {     // the incoming cell-state gradient enters through the in/out buffer
    hipError_t e = hipMemcpyAsync(grad_c, grad_cy, (size_t)n * 32 * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { spg_set_error("hipMemcpyAsync: %s", hipGetErrorString(e)); return (int)e; }
    p.use_dcdir = 1;
  }
  SPG_TRY(spg_launch_ecc_step_bwd(p, st));
  auto ident = [](const float* X, long ld) { SpgOperand o; memset(&o, 0, sizeof(o)); o.mode = SPG_PRO_IDENT; o.X = X; o.ld = ld; return o; };
  SpgWgradParams w; memset(&w, 0, sizeof(w));
  w.M = n; w.N = 128; w.K = 32;
  if (grads[0]) { w.a = ident(dgi, 128); w.b = ident(xg, 32); SPG_TRY(spg_launch_wgrad(w, grads[0], work, st)); }
  if (grads[1]) { w.a = ident(dgh, 128); w.b = ident(h, 32); SPG_TRY(spg_launch_wgrad(w, grads[1], work, st)); }
  if (grads[2]) SPG_TRY(spg_launch_colsum(dgi, 128, n, 128, grads[2], work, st));
  if (grads[3]) SPG_TRY(spg_launch_colsum(dgh, 128, n, 128, grads[3], work, st));
  if (ingate) {
    w.N = 32;
    if (grads[4]) { w.a = ident(dpre, 32); w.b = ident(h, 32); SPG_TRY(spg_launch_wgrad(w, grads[4], work, st)); }
    if (grads[5]) SPG_TRY(spg_launch_colsum(dpre, 32, n, 32, grads[5], work, st));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// fused dense layer
// ---------------------------------------------------------------------------------------------
static SpgOperand affine_operand(const float* X, long ld, int K, const float* sc, const float* sh, int relu) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  if (sc == nullptr && !relu) { o.mode = SPG_PRO_IDENT; }
  else { o.mode = SPG_PRO_AFFINE; o.c0 = sc; o.c1 = sh; o.relu = relu; o.n_affine = K; }
  o.X = X; o.ld = ld;
  return o;
}

extern "C" int spg_linear_fwd(const float* X, long ldx, int M, int K, const float* W, const float* bias, int N,
                              const float* in_scale, const float* in_shift, int in_relu, float* Y, long ldy, void* stream) {
  SPG_CHECK_ARG(X && W && Y, "null pointer");
  SPG_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale and in_shift go together");
  SpgGemmParams g; memset(&g, 0, sizeof(g));
  g.a = affine_operand(X, ldx, K, in_scale, in_shift, in_relu);
  g.W = W; g.ldw = K; g.bias = bias; g.M = M; g.N = N; g.K = K; g.rows_per_tile = M <= 8192 ? SPG_FC_ROWS : 128; g.epi = SPG_EPI_FWD; g.Y = Y; g.ldy = ldy;
  return spg_launch_gemm(g, (hipStream_t)stream);
}

// dX[M,K] = dY[M,N] @ W[N,K]   (data gradient of Y = X W^T; W is read untransposed)
extern "C" int spg_linear_dgrad(const float* dY, long lddy, int M, int N, const float* W, int K, float* dX, long ldx, void* stream) {
  SPG_CHECK_ARG(dY && W && dX, "null pointer");
  SpgGemmParams g; memset(&g, 0, sizeof(g));
  g.a = affine_operand(dY, lddy, N, nullptr, nullptr, 0);
  g.W = W; g.ldw = K; g.w_red = 1; g.M = M; g.N = K; g.K = N; g.rows_per_tile = M <= 8192 ? SPG_FC_ROWS : 128;
  g.epi = SPG_EPI_BWD; g.Y = dX; g.ldy = ldx; g.n_mask = K;
  return spg_launch_gemm(g, (hipStream_t)stream);
}

// out[N] = column sums of X[M,N] (bias gradient); work: >= 64 * N floats
extern "C" int spg_colsum(const float* X, long ldx, long M, int N, float* out, float* work, void* stream) {
  SPG_CHECK_ARG(X && out && work && M > 0 && N > 0, "bad argument");
  return spg_launch_colsum(X, ldx, M, N, out, work, (hipStream_t)stream);
}

extern "C" size_t spg_linear_wgrad_work_floats(int M, int N, int K) { return spg_wgrad_workspace_floats(M, N, K); }

extern "C" int spg_linear_wgrad(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K,
                                const float* in_scale, const float* in_shift, int in_relu, float* dW, float* work,
                                void* stream) {
  SPG_CHECK_ARG(dY && X && dW && work, "null pointer");
  SpgWgradParams w; memset(&w, 0, sizeof(w));
  w.a = affine_operand(dY, lddy, N, nullptr, nullptr, 0);
  w.b = affine_operand(X, ldx, K, in_scale, in_shift, in_relu);
  w.M = M; w.N = N; w.K = K;
  return spg_launch_wgrad(w, dW, work, (hipStream_t)stream);
}

// weight AND bias gradient of a dense layer in two launches (weight-gradient partials with the column sums of dY riding
// along, one batched reduction) instead of four (spg_linear_wgrad + spg_colsum)
extern "C" size_t spg_linear_wgrad_bias_work_floats(int M, int N, int K) {
  return ((spg_wgrad_workspace_floats(M, N, K) + 63) & ~(size_t)63) + ((spg_wgrad_colsum_floats(M, N, K) + 63) & ~(size_t)63) + 128;
}

extern "C" int spg_linear_wgrad_bias(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K,
                                     const float* in_scale, const float* in_shift, int in_relu, float* dW, float* dbias,
                                     float* work, void* stream) {
  SPG_CHECK_ARG(dY && X && dW && dbias && work, "null pointer");
  SpgWgradParams w; memset(&w, 0, sizeof(w));
  w.a = affine_operand(dY, lddy, N, nullptr, nullptr, 0);
  w.b = affine_operand(X, ldx, K, in_scale, in_shift, in_relu);
  w.M = M; w.N = N; w.K = K;
  SpgReduceQueue rq;
  rq.arena = work; rq.arena_floats = spg_linear_wgrad_bias_work_floats(M, N, K);
  SPG_TRY(spg_queue_wgrad(rq, w, dW, (hipStream_t)stream, dbias));
  return spg_flush_reduce(rq, (hipStream_t)stream);
}

// the whole backward of a dense layer Y = X W^T + b: data gradient, weight gradient and bias gradient are mutually
// independent -- ONE grouped launch (spg_gemm.h) + the batched reduction of the split partials, instead of three launches.
// dX may be null (the input needs no gradient); dbias may be null.  work: >= spg_linear_wgrad_bias_work_floats floats.
static int linear_backward_impl(const float* dY, long lddy, const float* X, long ldx, const float* W, int M, int N, int K, float* dX,
                                long lddx, float* dW, float* dbias, float* work, hipStream_t st, bool defer) {
  SPG_CHECK_ARG(dY && X && W && dW && work, "null pointer");
  SpgReduceQueue rq;
  rq.arena = work; rq.arena_floats = spg_linear_wgrad_bias_work_floats(M, N, K);
  {
    SpgGroupScope grp(st);
    SpgWgradParams w; memset(&w, 0, sizeof(w));
    w.a = affine_operand(dY, lddy, N, nullptr, nullptr, 0);
    w.b = affine_operand(X, ldx, K, nullptr, nullptr, 0);
    w.M = M; w.N = N; w.K = K;
    SPG_TRY(spg_queue_wgrad(rq, w, dW, st, dbias));
    if (dX != nullptr) SPG_TRY(spg_linear_dgrad(dY, lddy, M, N, W, K, dX, lddx, (void*)st));
    SPG_TRY(grp.flush());
  }
  if (defer) {      // the split partials are summed by the caller's next batched reduction (spg_gemm.h: spg_reduce_defer)
    for (int j = 0; j < rq.njobs; ++j) spg_reduce_defer(rq.jobs[j]);
    return 0;
  }
  return spg_flush_reduce(rq, st);
}

extern "C" int spg_linear_backward(const float* dY, long lddy, const float* X, long ldx, const float* W, int M, int N, int K,
                                   float* dX, long lddx, float* dW, float* dbias, float* work, void* stream) {
  return linear_backward_impl(dY, lddy, X, ldx, W, M, N, K, dX, lddx, dW, dbias, work, (hipStream_t)stream, false);
}

// inside spg_train_step, as part of a rider stage: weight + bias gradient only, as jobs of the group that is open on this
// thread (no scope of its own), the split partials left to the step's final batched reduction
int spg_linear_wgrad_bias_queue_deferred(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K, float* dW,
                                         float* dbias, float* work, hipStream_t st) {
  SPG_CHECK_ARG(dY && X && dW && work, "null pointer");
  SpgReduceQueue rq;
  rq.arena = work; rq.arena_floats = spg_linear_wgrad_bias_work_floats(M, N, K);
  SpgWgradParams w; memset(&w, 0, sizeof(w));
  w.a = affine_operand(dY, lddy, N, nullptr, nullptr, 0);
  w.b = affine_operand(X, ldx, K, nullptr, nullptr, 0);
  w.M = M; w.N = N; w.K = K;
  SPG_TRY(spg_queue_wgrad(rq, w, dW, st, dbias));
  for (int j = 0; j < rq.njobs; ++j) spg_reduce_defer(rq.jobs[j]);
  return 0;
}

// inside spg_train_step: the reduction of the partials waits for the step's final batched reduction
int spg_linear_backward_deferred(const float* dY, long lddy, const float* X, long ldx, const float* W, int M, int N, int K, float* dX,
                                 long lddx, float* dW, float* dbias, float* work, hipStream_t st) {
  return linear_backward_impl(dY, lddy, X, ldx, W, M, N, K, dX, lddx, dW, dbias, work, st, true);
}

// ---------------------------------------------------------------------------------------------
// element-wise gradient clamp + Adam on one flat buffer (learning/main.py:210-213, torch.optim.Adam semantics:
// weight decay added to the clamped gradient, bias-corrected moments, denom = sqrt(v)/sqrt(1-b2^t) + eps)
// ---------------------------------------------------------------------------------------------
__global__ void spg_adam_clamp_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                      float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                      float clip, float bc1, float bc2_sqrt, const float* __restrict__ grad_div,
                                      unsigned* __restrict__ guard, const float* __restrict__ guard_all) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // Fail-safe (round 6): guard[0] = the persistent RNN-ECC launches' time-out count of this device.  Non-zero means a wave of the
  // step whose gradients lie in `g` gave up waiting for a neighbour and went on with stale states: the gradients are WRONG, so the
  // update is withheld -- parameters and moments stay bit-identical -- and guard[1] counts the withheld launch.  The word is sticky
  // (only the host clears it: spg_ecc_persistent_status), so every later update is withheld too until the host has looked; no
  // host synchronisation is involved.  One scalar load per wave.
  // guard_all (data parallel): the ranks' time-out flags summed by the gradient all-reduce itself (FlatParameters.allreduce_sums) --
  // the summed gradients contain the failed rank's wrong ones, so EVERY rank withholds, and the replicas stay identical.
  const bool local_bad = guard != nullptr && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  const bool any_bad = guard_all != nullptr && *guard_all != 0.f;
  if (local_bad || any_bad) {
    if (i == 0 && guard != nullptr) atomicAdd(guard + 1, 1u);
    return;
  }
  if (i >= n) return;
  float gi = g[i];
  if (grad_div != nullptr) gi = gi / *grad_div;     // data-parallel normaliser (sum of the ranks' loss weights), still on the device
  if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
  g[i] = gi;                                   // the clamp is observable in p.grad, as in the reference loop
  if (wd != 0.f) gi = fmaf(wd, p[i], gi);
  const float mi = m[i] + (1.f - b1) * (gi - m[i]);          // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

extern "C" int spg_adam_clamp_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                           float beta1, float beta2, float eps, float weight_decay, float grad_clip, int step,
                                           const float* grad_div, const float* guard_all, void* stream) {
  SPG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad argument");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  const bool off = spg_tune_get(SPG_TUNE_NO_ADAM_GUARD) != 0;
  unsigned* guard = off ? nullptr : spg_px_guard_words();
  hipLaunchKernelGGL(spg_adam_clamp_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_clip, bc1, bc2_sqrt, grad_div, guard, off ? nullptr : guard_all);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_adam_clamp_step_scaled(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                          float beta1, float beta2, float eps, float weight_decay, float grad_clip, int step,
                                          const float* grad_div, void* stream) {
  return spg_adam_clamp_step_guarded(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_clip, step, grad_div, nullptr, stream);
}

// dst[0] = 1.0f if a one-launch recurrence of the current device has timed out since the host last cleared the word, else 0.0f --
// enqueued on `stream`; the data-parallel exchange puts it next to the gradients so that the all-reduce sums the ranks' flags
__global__ void spg_persistent_flag_kernel(const unsigned* __restrict__ guard, float* __restrict__ dst) {
  dst[0] = (guard != nullptr && __hip_atomic_load(const_cast<unsigned*>(guard), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.f : 0.f;
}
extern "C" int spg_ecc_persistent_flag(float* dst, void* stream) {
  SPG_CHECK_ARG(dst != nullptr, "null pointer");
  hipLaunchKernelGGL(spg_persistent_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, spg_px_guard_words(), dst);
  SPG_LAUNCH_CHECK();
  return 0;
}

extern "C" int spg_adam_clamp_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, float grad_clip, int step,
                                   void* stream) {
  return spg_adam_clamp_step_scaled(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_clip, step,
                                    nullptr, stream);
}
