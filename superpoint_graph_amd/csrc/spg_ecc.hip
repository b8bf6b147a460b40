// Edge-conditioned convolution + GRU kernels for gfx950 (reference: learning/ecc/GraphConvModule.py:44-152,
// learning/ecc/cuda_kernels.py:55-139, learning/modules.py:152-183,224-251).
//
// Design (MI355X-first, not a translation of the reference's one-thread-per-channel CUDA kernels):
//   * one 64-lane wavefront owns one graph node; the whole RNN-ECC iteration for that node
//     (gather x_src, per-edge 32x32 filter product, mean over the in-edges, input gate, the two
//     32->96 GRU projections, row normalisation, gates) is ONE kernel: the per-edge products and the
//     aggregate never exist in HBM;
//   * a 32x32 filter is read as 4 x 16-byte loads per lane (1 KiB per wave instruction, fully
//     coalesced); partial sums are reduced across lanes with wave shuffles once per node;
//   * no atomics anywhere: the backward walks a reverse CSR (by source) built once per batch, so
//     results are deterministic.
#include "spg_ecc.h"

// ---------------------------------------------------------------------------------------------
// graph build
// ---------------------------------------------------------------------------------------------
static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

struct GraphLayout {
  size_t hdr, rowptr, src, dst, rev_eid, invdeg, rev_rowptr, cursor, total;
};
// target-side arrays first (their offsets depend on N and E only); the source-side arrays, whose length is
// the number of SOURCE rows Ns >= N (GraphConvFunction allows more input rows than output nodes), come last.
static GraphLayout graph_layout(int N, int Ns, int E) {
  GraphLayout L;
  size_t o = 0;
  L.hdr = o; o += 16;
  L.rowptr = o; o += al16((size_t)(N + 1) * 4);
  L.src = o; o += al16((size_t)(E + 1) * 4);
  L.dst = o; o += al16((size_t)(E + 1) * 4);
  L.rev_eid = o; o += al16((size_t)(E + 1) * 4);
  L.invdeg = o; o += al16((size_t)(N + 1) * 4);
  L.rev_rowptr = o; o += al16((size_t)(Ns + 1) * 4);
  L.cursor = o; o += al16((size_t)(Ns + 1) * 4);
  L.total = o;
  return L;
}
size_t spg_graph_bytes(int N, int Ns, int E) { return graph_layout(N, Ns < N ? N : Ns, E).total; }

SpgGraph spg_graph_view(const void* ws, int N, int E) {
  GraphLayout L = graph_layout(N, N, E);
  const char* b = (const char*)ws;
  SpgGraph g;
  g.N = N; g.E = E;
  g.hdr = (const int*)(b + L.hdr);
  g.rowptr = (const int*)(b + L.rowptr);
  g.src = (const int*)(b + L.src);
  g.dst = (const int*)(b + L.dst);
  g.rev_rowptr = (const int*)(b + L.rev_rowptr);
  g.rev_eid = (const int*)(b + L.rev_eid);
  g.invdeg = (const float*)(b + L.invdeg);
  return g;
}

template <typename TIn>
__global__ __launch_bounds__(1024) void spg_scan_kernel(const TIn* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ long part[1024];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int b = min(n, t * chunk), e = min(n, b + chunk);
  long s = 0;
  for (int i = b; i < e; ++i) s += (long)in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const long v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long base = part[t] - s;
  for (int i = b; i < e; ++i) {
    out[i] = (int)base;
    base += (long)in[i];
  }
  if (t == 1023) out[n] = (int)part[1023];
}

__global__ void spg_graph_nodes_kernel(const int64_t* __restrict__ degs, const int* __restrict__ rowptr, int N, int Ns,
                                       int E, int* __restrict__ hdr, int* __restrict__ dst, float* __restrict__ invdeg,
                                       int* __restrict__ cursor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { hdr[0] = N; hdr[1] = Ns; hdr[2] = E; hdr[3] = 0; }
  if (i < Ns) cursor[i] = 0;
  if (i >= N) return;
  const int d = (int)degs[i];
  invdeg[i] = d > 0 ? 1.0f / (float)d : 0.f;
  const int e0 = rowptr[i];
  for (int k = 0; k < d; ++k) dst[e0 + k] = i;
}

__global__ void spg_graph_edges_kernel(const int64_t* __restrict__ idxn, int E, int Ns, int* __restrict__ src,
                                       int* __restrict__ cnt, int* __restrict__ hdr) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t j64 = idxn[e];
  if (j64 < 0 || j64 >= Ns) {   // malformed index buffer: flag it, never write out of bounds
    src[e] = 0;
    atomicOr(&hdr[3], 1);
    return;
  }
  const int j = (int)j64;
  src[e] = j;
  atomicAdd(&cnt[j], 1);   // integer count only: the result is order-independent
}

__global__ void spg_graph_revfill_kernel(const int* __restrict__ src, int E, const int* __restrict__ rev_rowptr,
                                         int* __restrict__ cursor, int* __restrict__ rev_eid) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int j = src[e];
  const int pos = atomicAdd(&cursor[j], 1);
  rev_eid[rev_rowptr[j] + pos] = e;
}

__global__ void spg_graph_revsort_kernel(const int* __restrict__ rev_rowptr, int N, int* __restrict__ rev_eid) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int b = rev_rowptr[j], e = rev_rowptr[j + 1];
  for (int a = b + 1; a < e; ++a) {   // insertion sort: out-degrees are small
    const int v = rev_eid[a];
    int k = a - 1;
    while (k >= b && rev_eid[k] > v) { rev_eid[k + 1] = rev_eid[k]; --k; }
    rev_eid[k + 1] = v;
  }
}

int spg_graph_build_impl(const int64_t* idxn, const int64_t* degs, int N, int Ns, int E, void* ws, hipStream_t stream) {
  SPG_CHECK_ARG(N > 0 && E >= 0, "graph needs N > 0");
  if (Ns < N) Ns = N;
  GraphLayout L = graph_layout(N, Ns, E);
  char* b = (char*)ws;
  int* rowptr = (int*)(b + L.rowptr);
  int* src = (int*)(b + L.src);
  int* dst = (int*)(b + L.dst);
  int* rev_rowptr = (int*)(b + L.rev_rowptr);
  int* rev_eid = (int*)(b + L.rev_eid);
  float* invdeg = (float*)(b + L.invdeg);
  int* cursor = (int*)(b + L.cursor);
  int* hdr = (int*)(b + L.hdr);
  hipLaunchKernelGGL(spg_scan_kernel<int64_t>, dim3(1), dim3(1024), 0, stream, degs, N, rowptr);
  SPG_LAUNCH_CHECK();
  hipLaunchKernelGGL(spg_graph_nodes_kernel, dim3(spg_cdiv(Ns, 256)), dim3(256), 0, stream, degs, rowptr, N, Ns, E, hdr, dst,
                     invdeg, cursor);
  SPG_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(spg_graph_edges_kernel, dim3(spg_cdiv(E, 256)), dim3(256), 0, stream, idxn, E, Ns, src, cursor, hdr);
    SPG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(spg_scan_kernel<int>, dim3(1), dim3(1024), 0, stream, (const int*)cursor, Ns, rev_rowptr);
  SPG_LAUNCH_CHECK();
  hipError_t me = hipMemsetAsync(cursor, 0, (size_t)Ns * 4, stream);
  if (me != hipSuccess) { spg_set_error("hipMemsetAsync failed: %s", hipGetErrorString(me)); return (int)me; }
  if (E > 0) {
    hipLaunchKernelGGL(spg_graph_revfill_kernel, dim3(spg_cdiv(E, 256)), dim3(256), 0, stream, (const int*)src, E,
                       (const int*)rev_rowptr, cursor, rev_eid);
    SPG_LAUNCH_CHECK();
    hipLaunchKernelGGL(spg_graph_revsort_kernel, dim3(spg_cdiv(Ns, 256)), dim3(256), 0, stream, (const int*)rev_rowptr, Ns,
                       rev_eid);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float spg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

#define SPG_IN_EPS 1e-5f   // nn.InstanceNorm1d(1, eps=1e-5), learning/modules.py:213-214

// in-edge aggregation for node i; result: every lane of the wave holds, for the matrix mode, the 4
// output channels 4*(lane&7)..+3 in a4[]; for the vector mode lanes 0..31 hold channel `lane` in a4[0].
__device__ __forceinline__ void spg_aggregate_node(const SpgGraph& g, const float* __restrict__ W, int matrix,
                                                   const float* __restrict__ hin, long ld, int i, int lane,
                                                   float (&a4)[4]) {
  a4[0] = a4[1] = a4[2] = a4[3] = 0.f;
  const int e0 = g.rowptr[i], e1 = g.rowptr[i + 1];
  if (matrix) {
    const int kb = lane >> 3;
    // 4 edges per batch: all index / filter / state loads of a batch are issued before any of them is consumed
    for (int e = e0; e < e1; e += 4) {
      int sidx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sidx[u] = g.src[min(e + u, e1 - 1)];
      f32x4 w[4][4];
      float xk[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4* We = reinterpret_cast<const f32x4*>(W + (long)min(e + u, e1 - 1) * 1024);
        const float* xj = hin + (long)sidx[u] * ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w[u][q] = We[lane + 64 * q];          // row k = kb + 8q, columns 4*(lane&7)..+3 of W_e[in][out]
          xk[u][q] = xj[kb + 8 * q];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float on = (e + u < e1) ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xv = xk[u][q] * on;
          a4[0] = fmaf(xv, w[u][q][0], a4[0]);
          a4[1] = fmaf(xv, w[u][q][1], a4[1]);
          a4[2] = fmaf(xv, w[u][q][2], a4[2]);
          a4[3] = fmaf(xv, w[u][q][3], a4[3]);
        }
      }
    }
#pragma unroll
    for (int off = 8; off <= 32; off <<= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) a4[c] += __shfl_xor(a4[c], off, 64);
    }
    const float s = g.invdeg[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) a4[c] *= s;
  } else {
    if (lane < 32) {
      float a = 0.f;
      for (int e = e0; e < e1; ++e) a = fmaf(hin[(long)g.src[e] * ld + lane], W[(long)e * 32 + lane], a);
      a4[0] = a * g.invdeg[i];
    }
  }
}

// GRUCellEx forward internals for one node, lane layout:
//   first value  (lanes 0..63): gate pre-activation index `lane`      (0..31 reset chunk, 32..63 update chunk)
//   second value (lanes 0..31): gate pre-activation index 64 + lane   (new-gate chunk)
struct GruFwdState {
  float gin, x;          // input gate and gated input (lanes 0..31)
  float ui1, ui2;        // normalised W_ih x
  float uh1, uh2;        // normalised W_hh h
  float rstd_i, rstd_h;
  float r, z, n;         // gates (lanes 0..31; z was shuffled from lanes 32..63)
};

// Gate rows of the cell's weight matrices, read from a workgroup copy in LDS, rows padded to 33 floats (conflict-free both
// along a row -- the forward dot products -- and along a column -- the backward's W^T products): lane l owns gate rows l
// (first value) and 64 + (l & 31) (GRU) / 64 + l (LSTM) (second value) of W_ih / W_hh and row (l & 31) of the input gate.
#define SPG_WLD 33
struct GruRowsLds {
  const float *ih1, *hh1, *ih2, *hh2, *ig;     // row pointers of this lane
  __device__ __forceinline__ float dot(const float* __restrict__ row, const float* __restrict__ v) const {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(v + 4 * q);   // LDS broadcast read
      a0 = fmaf(row[4 * q + 0], x[0], a0); a1 = fmaf(row[4 * q + 1], x[1], a1);
      a0 = fmaf(row[4 * q + 2], x[2], a0); a1 = fmaf(row[4 * q + 3], x[3], a1);
    }
    return a0 + a1;
  }
  __device__ __forceinline__ float dot_ih1(const float* v) const { return dot(ih1, v); }
  __device__ __forceinline__ float dot_hh1(const float* v) const { return dot(hh1, v); }
  __device__ __forceinline__ float dot_ih2(const float* v) const { return dot(ih2, v); }
  __device__ __forceinline__ float dot_hh2(const float* v) const { return dot(hh2, v); }
  __device__ __forceinline__ float dot_ig(const float* v) const { return dot(ig, v); }
};

// block-wide: global [GW][32] x2 + [32][32] -> LDS rows of SPG_WLD floats (w_ih | w_hh | w_ig); caller synchronises
template <int GW>
__device__ __forceinline__ void spg_stage_cell_weights(const SpgGruParams& G, float* __restrict__ sw) {
  constexpr int NQ = (2 * GW + 32) * 8;          // float4 pieces
  for (int idx = threadIdx.x; idx < NQ; idx += SPG_THREADS) {
    const int row = idx >> 3, q = idx & 7;
    const f32x4* src = row < GW ? reinterpret_cast<const f32x4*>(G.w_ih) + row * 8
                     : (row < 2 * GW ? reinterpret_cast<const f32x4*>(G.w_hh) + (row - GW) * 8
                                     : reinterpret_cast<const f32x4*>(G.ingate ? G.w_ig : G.w_ih) + (row - 2 * GW) * 8);
    const f32x4 v = src[q];
    float* d = sw + row * SPG_WLD + 4 * q;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}
template <int CELL>
__device__ __forceinline__ void spg_gru_lds_rows(const float* __restrict__ sw, int lane, GruRowsLds& w) {
  constexpr int GW = CELL == SPG_CELL_LSTM ? 128 : 96;
  const int r1 = lane, r2 = CELL == SPG_CELL_LSTM ? 64 + lane : 64 + (lane & 31);
  w.ih1 = sw + r1 * SPG_WLD; w.ih2 = sw + r2 * SPG_WLD;
  w.hh1 = sw + (GW + r1) * SPG_WLD; w.hh2 = sw + (GW + r2) * SPG_WLD;
  w.ig = sw + (2 * GW + (lane & 31)) * SPG_WLD;
}

template <class Rows>
__device__ __forceinline__ void spg_gru_forward_node(const SpgGruParams& G, const Rows& w, const float* __restrict__ sa,
                                                     const float* __restrict__ sh, float* __restrict__ sx, int lane,
                                                     GruFwdState& st) {
  // input gate: x = sigmoid(W_ig h + b_ig) * a      (learning/modules.py:225-226)
  float gin = 1.f, x = 0.f;
  if (lane < 32) {
    if (G.ingate) gin = spg_sigmoid(w.dot_ig(sh) + G.b_ig[lane]);
    x = gin * sa[lane];
    sx[lane] = x;
  }
  st.gin = gin; st.x = x;
  __syncthreads();
  float gi1 = w.dot_ih1(sx), gh1 = w.dot_hh1(sh);
  float gi2 = 0.f, gh2 = 0.f;
  if (lane < 32) { gi2 = w.dot_ih2(sx); gh2 = w.dot_hh2(sh); }
  st.rstd_i = 1.f; st.rstd_h = 1.f;
  if (G.layernorm) {   // per-row (x - mean)/sqrt(var_biased + eps) over the 96 values (learning/modules.py:218-222)
    const float mi = spg_wave_sum(gi1 + (lane < 32 ? gi2 : 0.f)) * (1.f / 96.f);
    const float mh = spg_wave_sum(gh1 + (lane < 32 ? gh2 : 0.f)) * (1.f / 96.f);
    const float di1 = gi1 - mi, di2 = gi2 - mi, dh1 = gh1 - mh, dh2 = gh2 - mh;
    const float vi = spg_wave_sum(di1 * di1 + (lane < 32 ? di2 * di2 : 0.f)) * (1.f / 96.f);
    const float vh = spg_wave_sum(dh1 * dh1 + (lane < 32 ? dh2 * dh2 : 0.f)) * (1.f / 96.f);
    st.rstd_i = 1.0f / sqrtf(vi + SPG_IN_EPS);
    st.rstd_h = 1.0f / sqrtf(vh + SPG_IN_EPS);
    gi1 = di1 * st.rstd_i; gi2 = di2 * st.rstd_i; gh1 = dh1 * st.rstd_h; gh2 = dh2 * st.rstd_h;
  }
  st.ui1 = gi1; st.ui2 = gi2; st.uh1 = gh1; st.uh2 = gh2;
  // gates (learning/modules.py:242-250): first values are r (lanes 0..31) and z (lanes 32..63)
  const float g1 = spg_sigmoid(((gi1 + G.b_ih[lane]) + gh1) + G.b_hh[lane]);
  st.r = g1;
  st.z = __shfl(g1, (lane & 31) + 32, 64);
  st.n = 0.f;
  if (lane < 32) st.n = tanhf((gi2 + G.b_ih[64 + lane]) + g1 * (gh2 + G.b_hh[64 + lane]));
}


// LSTMCellEx forward internals for one node (learning/modules.py:280-309).  The 128 gate pre-activations
// (chunks i | f | g | o of 32) are two values per lane: index `lane` (i on lanes 0..31, f on 32..63) and index
// 64 + lane (g on lanes 0..31, o on 32..63).  Unlike the GRU, the biases are added BEFORE the row normalisation
// (nnf.linear(input, weight_ih, bias_ih), :297-299).
struct LstmFwdState {
  float gin, x;          // input gate and gated input (lanes 0..31)
  float ui1, ui2;        // normalised W_ih x + b_ih
  float uh1, uh2;        // normalised W_hh h + b_hh
  float rstd_i, rstd_h;
  float i, f, g, o;      // gates, channel = lane (lanes 0..31)
  float c_prev, tc, cy, hy;
};

template <class Rows>
__device__ __forceinline__ void spg_lstm_forward_node(const SpgGruParams& G, const Rows& w, const float* __restrict__ sa,
                                                      const float* __restrict__ sh, float* __restrict__ sx, float c_prev,
                                                      int lane, LstmFwdState& st) {
  float gin = 1.f, x = 0.f;
  if (lane < 32) {
    if (G.ingate) gin = spg_sigmoid(w.dot_ig(sh) + G.b_ig[lane]);      // :285-286, hidden[0]
    x = gin * sa[lane];
    sx[lane] = x;
  }
  st.gin = gin; st.x = x;
  __syncthreads();
  float gi1 = w.dot_ih1(sx) + G.b_ih[lane], gi2 = w.dot_ih2(sx) + G.b_ih[64 + lane];
  float gh1 = w.dot_hh1(sh) + G.b_hh[lane], gh2 = w.dot_hh2(sh) + G.b_hh[64 + lane];
  st.rstd_i = 1.f; st.rstd_h = 1.f;
  if (G.layernorm) {   // InstanceNorm1d over the 128 values of the row, biased variance (:275-279)
    const float mi = spg_wave_sum(gi1 + gi2) * (1.f / 128.f);
    const float mh = spg_wave_sum(gh1 + gh2) * (1.f / 128.f);
    const float di1 = gi1 - mi, di2 = gi2 - mi, dh1 = gh1 - mh, dh2 = gh2 - mh;
    const float vi = spg_wave_sum(di1 * di1 + di2 * di2) * (1.f / 128.f);
    const float vh = spg_wave_sum(dh1 * dh1 + dh2 * dh2) * (1.f / 128.f);
    st.rstd_i = 1.0f / sqrtf(vi + SPG_IN_EPS);
    st.rstd_h = 1.0f / sqrtf(vh + SPG_IN_EPS);
    gi1 = di1 * st.rstd_i; gi2 = di2 * st.rstd_i; gh1 = dh1 * st.rstd_h; gh2 = dh2 * st.rstd_h;
  }
  st.ui1 = gi1; st.ui2 = gi2; st.uh1 = gh1; st.uh2 = gh2;
  const float p1 = gi1 + gh1, p2 = gi2 + gh2;
  const float v1 = spg_sigmoid(p1);                         // i (lanes 0..31) / f (lanes 32..63)
  const float v2 = lane < 32 ? tanhf(p2) : spg_sigmoid(p2); // g (lanes 0..31) / o (lanes 32..63)
  st.i = v1; st.g = v2;
  st.f = __shfl(v1, (lane & 31) + 32, 64);
  st.o = __shfl(v2, (lane & 31) + 32, 64);
  st.c_prev = c_prev;
  st.cy = st.f * c_prev + st.i * st.g;                      // :306
  st.tc = tanhf(st.cy);
  st.hy = st.o * st.tc;                                     // :307
}

// ---------------------------------------------------------------------------------------------
// forward step
// ---------------------------------------------------------------------------------------------
// The cell's weight matrices are staged once per workgroup in LDS (rows padded to SPG_WLD floats) and every lane reads its
// gate rows from there: 103-168 VGPRs, 3-4 waves per SIMD.  (Register-resident rows -- 264 VGPRs, one wave per SIMD -- were
// measured slower at every graph size: 1 scene 1.83 -> 1.79 ms/step, 8 scenes 811k -> 887k superpoints/s.)
template <int CELL>
__global__ __launch_bounds__(256, 3) void spg_ecc_step_fwd_kernel(const SpgEccStepFwd p) {
  constexpr int GW = CELL == SPG_CELL_LSTM ? 128 : 96;
  __shared__ __attribute__((aligned(16))) float lds[4][3][32];
  __shared__ float sw[(2 * GW + 32) * SPG_WLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  const bool active = i < p.g.N;
  GruRowsLds wr;
  if (p.do_gru) spg_stage_cell_weights<GW>(p.gru, sw);          // visible after the barrier behind the aggregation
  spg_gru_lds_rows<CELL>(sw, lane, wr);
  float* sa = lds[wave][0];
  float* sh = lds[wave][1];
  float* sx = lds[wave][2];
  if (active) {
    if (p.agg_in != nullptr) {
      if (lane < 32) sa[lane] = p.agg_in[(long)i * p.ldagg + lane];
    } else {
      float a4[4];
      spg_aggregate_node(p.g, p.W, p.matrix, p.hin, p.ld, i, lane, a4);
      if (p.matrix) {
        if (lane < 8) {
          sa[4 * lane + 0] = a4[0]; sa[4 * lane + 1] = a4[1]; sa[4 * lane + 2] = a4[2]; sa[4 * lane + 3] = a4[3];
        }
      } else if (lane < 32) {
        sa[lane] = a4[0];
      }
    }
    if (lane < 32 && p.do_gru) sh[lane] = p.hin[(long)i * p.ld + lane];
  } else if (lane < 32) {
    sa[lane] = 0.f; sh[lane] = 0.f;
  }
  __syncthreads();
  if (active && p.agg_save != nullptr && lane < 32) p.agg_save[(long)i * p.ldagg + lane] = sa[lane];
  if (!p.do_gru) return;
  if constexpr (CELL == SPG_CELL_GRU) {
    GruFwdState st;
    spg_gru_forward_node(p.gru, wr, sa, sh, sx, lane, st);
    if (active && lane < 32) {
      const float h = sh[lane];
      p.hout[(long)i * p.ld + lane] = st.n + st.z * (h - st.n);   // hy = newgate + inputgate*(hidden - newgate)
    }
  } else {
    LstmFwdState st;
    const float c = (active && lane < 32 && p.cin != nullptr) ? p.cin[(long)i * p.ld + lane] : 0.f;
    spg_lstm_forward_node(p.gru, wr, sa, sh, sx, c, lane, st);
    if (active && lane < 32) {
      p.hout[(long)i * p.ld + lane] = st.hy;
      p.cout[(long)i * p.ld + lane] = st.cy;
    }
  }
}

int spg_launch_ecc_step_fwd(const SpgEccStepFwd& p, hipStream_t stream) {
  const dim3 grid(spg_cdiv(p.g.N, 4));
  if (p.cell == SPG_CELL_LSTM) hipLaunchKernelGGL(spg_ecc_step_fwd_kernel<SPG_CELL_LSTM>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(spg_ecc_step_fwd_kernel<SPG_CELL_GRU>, grid, dim3(256), 0, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// backward step
// ---------------------------------------------------------------------------------------------
template <int CELL>
__global__ __launch_bounds__(256, 3) void spg_ecc_step_bwd_kernel(const SpgEccStepBwd p) {
  constexpr int GW = CELL == SPG_CELL_LSTM ? 128 : 96;
  __shared__ __attribute__((aligned(16))) float lds[4][5][GW];
  // the backward needs COLUMNS of the weight matrices (dx[c] = sum_o W[o][c] dg[o]): the block stages them in LDS once
  // (loads issued here, in flight during the reverse gather) instead of 2*GW+32 global loads per lane
  constexpr int WQ = (2 * GW * 32 + 1024) / 4;                   // float4 pieces: w_ih | w_hh | w_ig
  __shared__ float sw[(2 * GW + 32) * SPG_WLD];                   // rows padded to SPG_WLD: conflict-free along rows and columns
  constexpr int WPT = (WQ + SPG_THREADS - 1) / SPG_THREADS;
  f32x4 wreg[WPT];
  if (!p.final_only) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int idx = threadIdx.x + SPG_THREADS * u;
      const int ii = idx < WQ ? idx : 0;
      const f32x4* src = ii < GW * 8 ? reinterpret_cast<const f32x4*>(p.gru.w_ih) + ii
                       : (ii < 2 * GW * 8 ? reinterpret_cast<const f32x4*>(p.gru.w_hh) + (ii - GW * 8)
                                          : reinterpret_cast<const f32x4*>(p.gru.ingate ? p.gru.w_ig : p.gru.w_ih) + (ii - 2 * GW * 8));
      wreg[u] = *src;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  const bool active = j < p.g.N;
  float* sa = lds[wave][0];        // [32] aggregate   | later dgi [96]
  float* sh = lds[wave][1];        // [32] hidden      | later dgh [96]
  float* sx = lds[wave][2];        // [32] gated input
  float* sd = lds[wave][3];        // [32] dH          | later dpre [32]
  // ---- phase 1: total gradient wrt the state produced by this iteration ----
  float dH = 0.f;   // lanes 0..31
  if (active) {
    if (lane < 32) {
      if (p.dcat != nullptr) dH += p.dcat[(long)j * p.ldc + lane];
      if (p.use_dhdir) dH += p.dhdir[(long)j * 32 + lane];
    }
    if (p.Gnext != nullptr) {
      const int b = p.g.rev_rowptr[j], e_ = p.g.rev_rowptr[j + 1];
      if (p.matrix) {
        float pq[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = b; t < e_; t += 4) {      // 4 out-edges per batch, loads first
          int eid[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) eid[u] = p.g.rev_eid[min(t + u, e_ - 1)];
          int did[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) did[u] = p.g.dst[eid[u]];
          f32x4 w[4][4], g4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            g4[u] = *reinterpret_cast<const f32x4*>(p.Gnext + (long)did[u] * p.ldg + 4 * (lane & 7));
            const f32x4* We = reinterpret_cast<const f32x4*>(p.W + (long)eid[u] * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[u][q] = We[lane + 64 * q];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float on = (t + u < e_) ? 1.f : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              pq[q] += on * ((w[u][q][0] * g4[u][0] + w[u][q][1] * g4[u][1]) + (w[u][q][2] * g4[u][2] + w[u][q][3] * g4[u][3]));
          }
        }
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pq[q] += __shfl_xor(pq[q], off, 64);
        }
        if ((lane & 7) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sd[(lane >> 3) + 8 * q] = pq[q];   // input channel k = lane/8 + 8q
        }
      } else if (lane < 32) {
        float s = 0.f;
        for (int t = b; t < e_; ++t) {
          const int e = p.g.rev_eid[t];
          s = fmaf(p.W[(long)e * 32 + lane], p.Gnext[(long)p.g.dst[e] * p.ldg + lane], s);
        }
        sd[lane] = s;
      }
    } else if (lane < 32) {
      sd[lane] = 0.f;
    }
  } else if (lane < 32) {
    sd[lane] = 0.f;
  }
  __syncthreads();
  if (lane < 32) dH += sd[lane];
  if (p.final_only) {
    if (active && lane < 32) p.gx[(long)j * 32 + lane] = dH;
    return;
  }
  // ---- phase 2: recompute the GRU forward of this iteration, then its backward ----
  if (lane < 32) {
    sa[lane] = active ? p.agg[(long)j * p.ldagg + lane] : 0.f;
    sh[lane] = active ? p.hin[(long)j * p.ld + lane] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < WPT; ++u) {
    const int idx = threadIdx.x + SPG_THREADS * u;
    if (idx < WQ) {
      float* d = sw + (idx >> 3) * SPG_WLD + 4 * (idx & 7);
      d[0] = wreg[u][0]; d[1] = wreg[u][1]; d[2] = wreg[u][2]; d[3] = wreg[u][3];
    }
  }
  const float* sw_ih = sw;
  const float* sw_hh = sw_ih + GW * SPG_WLD;
  const float* sw_ig = sw_hh + GW * SPG_WLD;
  __syncthreads();
  const SpgGruParams& G = p.gru;
  if constexpr (CELL == SPG_CELL_LSTM) {
    // ---- LSTMCellEx backward (learning/modules.py:280-309) ----
    LstmFwdState st;
    {
      const float c = (active && lane < 32 && p.cin != nullptr) ? p.cin[(long)j * p.ld + lane] : 0.f;
      GruRowsLds wr;
      spg_gru_lds_rows<SPG_CELL_LSTM>(sw, lane, wr);
      spg_lstm_forward_node(G, wr, sa, sh, sx, c, lane, st);
    }
    const float a_in = lane < 32 ? sa[lane] : 0.f;
    // gate backward on lanes 0..31 (channel = lane):  hy = o tanh(cy),  cy = f c + i g
    float di_pre = 0.f, df_pre = 0.f, dg_pre = 0.f, do_pre = 0.f, dc_prev = 0.f;
    if (lane < 32) {
      float dC = dH * st.o * (1.f - st.tc * st.tc);
      if (p.use_dcdir && active) dC += p.dcdir[(long)j * 32 + lane];
      do_pre = dH * st.tc * st.o * (1.f - st.o);
      di_pre = dC * st.g * st.i * (1.f - st.i);
      df_pre = dC * st.c_prev * st.f * (1.f - st.f);
      dg_pre = dC * st.i * (1.f - st.g * st.g);
      dc_prev = dC * st.f;
    }
    // gradient wrt the sum of the two normalised pre-activations, in the two-values-per-lane layout
    const float fsh = __shfl(df_pre, lane & 31, 64), osh = __shfl(do_pre, lane & 31, 64);
    const float du1 = lane < 32 ? di_pre : fsh;      // index lane
    const float du2 = lane < 32 ? dg_pre : osh;      // index 64 + lane
    float dgi1 = du1, dgi2 = du2, dgh1 = du1, dgh2 = du2;
    if (G.layernorm) {
      const float m1 = spg_wave_sum(du1 + du2) * (1.f / 128.f);
      const float m2i = spg_wave_sum(du1 * st.ui1 + du2 * st.ui2) * (1.f / 128.f);
      const float m2h = spg_wave_sum(du1 * st.uh1 + du2 * st.uh2) * (1.f / 128.f);
      dgi1 = st.rstd_i * (du1 - m1 - st.ui1 * m2i);
      dgi2 = st.rstd_i * (du2 - m1 - st.ui2 * m2i);
      dgh1 = st.rstd_h * (du1 - m1 - st.uh1 * m2h);
      dgh2 = st.rstd_h * (du2 - m1 - st.uh2 * m2h);
    }
    __syncthreads();   // sa/sh (as inputs) are no longer needed by any lane of this wave
    sa[lane] = dgi1; sa[64 + lane] = dgi2;
    sh[lane] = dgh1; sh[64 + lane] = dgh2;
    if (active) {
      p.dgi[(long)j * p.ld96 + lane] = dgi1; p.dgi[(long)j * p.ld96 + 64 + lane] = dgi2;
      p.dgh[(long)j * p.ld96 + lane] = dgh1; p.dgh[(long)j * p.ld96 + 64 + lane] = dgh2;
    }
    __syncthreads();
    float dx = 0.f, dh_acc = 0.f;
    if (lane < 32) {
#pragma unroll 8
      for (int o = 0; o < 128; ++o) {
        dx = fmaf(sw_ih[o * SPG_WLD + lane], sa[o], dx);
        dh_acc = fmaf(sw_hh[o * SPG_WLD + lane], sh[o], dh_acc);
      }
    }
    float da = dx, dpre = 0.f;
    if (G.ingate) {
      da = dx * st.gin;
      dpre = dx * a_in * st.gin * (1.f - st.gin);
      if (lane < 32) sd[lane] = dpre;
    }
    __syncthreads();
    if (G.ingate && lane < 32) {
#pragma unroll 8
      for (int o = 0; o < 32; ++o) dh_acc = fmaf(sw_ig[o * SPG_WLD + lane], sd[o], dh_acc);
    }
    if (active && lane < 32) {
      p.dpre[(long)j * p.ld32 + lane] = dpre;
      p.xg[(long)j * p.ld32 + lane] = st.x;
      p.dhdir[(long)j * 32 + lane] = dh_acc;
      p.dcdir[(long)j * 32 + lane] = dc_prev;
      p.Gcur[(long)j * p.ldg + lane] = p.g.invdeg != nullptr ? da * p.g.invdeg[j] : da;
    }
  } else {
    GruFwdState st;
    {
      GruRowsLds wr;
      spg_gru_lds_rows<SPG_CELL_GRU>(sw, lane, wr);
      spg_gru_forward_node(G, wr, sa, sh, sx, lane, st);
    }
    const float a_in = lane < 32 ? sa[lane] : 0.f;
    const float h_in = lane < 32 ? sh[lane] : 0.f;
    // gate backward on lanes 0..31 (channel = lane)
    float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dh_acc = 0.f;
    if (lane < 32) {
      const float dn = dH * (1.f - st.z);
      const float dzg = dH * (h_in - st.n);
      dh_acc = dH * st.z;
      dn_pre = dn * (1.f - st.n * st.n);
      const float dr = dn_pre * (st.uh2 + G.b_hh[64 + lane]);
      dz_pre = dzg * st.z * (1.f - st.z);
      dr_pre = dr * st.r * (1.f - st.r);
    }
    // gradients wrt the normalised pre-activations, in the first/second-value lane layout
    const float zsh = __shfl(dz_pre, lane & 31, 64);
    float dui1 = lane < 32 ? dr_pre : zsh;        // index lane
    float dui2 = lane < 32 ? dn_pre : 0.f;        // index 64 + lane
    float duh1 = dui1;
    float duh2 = lane < 32 ? dn_pre * st.r : 0.f;
    if (active) {
      p.dui[(long)j * p.ld96 + lane] = dui1;
      p.duh[(long)j * p.ld96 + lane] = duh1;
      if (lane < 32) {
        p.dui[(long)j * p.ld96 + 64 + lane] = dui2;
        p.duh[(long)j * p.ld96 + 64 + lane] = duh2;
      }
    }
    // through the row normalisation: dg = rstd * (du - mean(du) - u * mean(du*u))
    float dgi1 = dui1, dgi2 = dui2, dgh1 = duh1, dgh2 = duh2;
    if (G.layernorm) {
      const float m1i = spg_wave_sum(dui1 + dui2) * (1.f / 96.f);
      const float m2i = spg_wave_sum(dui1 * st.ui1 + dui2 * st.ui2) * (1.f / 96.f);
      const float m1h = spg_wave_sum(duh1 + duh2) * (1.f / 96.f);
      const float m2h = spg_wave_sum(duh1 * st.uh1 + duh2 * st.uh2) * (1.f / 96.f);
      dgi1 = st.rstd_i * (dui1 - m1i - st.ui1 * m2i);
      dgi2 = st.rstd_i * (dui2 - m1i - st.ui2 * m2i);
      dgh1 = st.rstd_h * (duh1 - m1h - st.uh1 * m2h);
      dgh2 = st.rstd_h * (duh2 - m1h - st.uh2 * m2h);
    }
    __syncthreads();   // sa/sh (as inputs) are no longer needed by any lane of this wave
    sa[lane] = dgi1;
    sh[lane] = dgh1;
    if (lane < 32) { sa[64 + lane] = dgi2; sh[64 + lane] = dgh2; }
    if (active) {
      p.dgi[(long)j * p.ld96 + lane] = dgi1;
      p.dgh[(long)j * p.ld96 + lane] = dgh1;
      if (lane < 32) {
        p.dgi[(long)j * p.ld96 + 64 + lane] = dgi2;
        p.dgh[(long)j * p.ld96 + 64 + lane] = dgh2;
      }
    }
    __syncthreads();
    float dx = 0.f;
    if (lane < 32) {
  #pragma unroll 8
      for (int o = 0; o < 96; ++o) {
        dx = fmaf(sw_ih[o * SPG_WLD + lane], sa[o], dx);
        dh_acc = fmaf(sw_hh[o * SPG_WLD + lane], sh[o], dh_acc);
      }
    }
    float da = dx, dpre = 0.f;
    if (G.ingate) {
      da = dx * st.gin;
      dpre = dx * a_in * st.gin * (1.f - st.gin);
      if (lane < 32) sd[lane] = dpre;
    }
    __syncthreads();
    if (G.ingate && lane < 32) {
  #pragma unroll 8
      for (int o = 0; o < 32; ++o) dh_acc = fmaf(sw_ig[o * SPG_WLD + lane], sd[o], dh_acc);
    }
    if (active && lane < 32) {
      p.dpre[(long)j * p.ld32 + lane] = dpre;
      p.xg[(long)j * p.ld32 + lane] = st.x;
      p.dhdir[(long)j * 32 + lane] = dh_acc;
      p.Gcur[(long)j * p.ldg + lane] = p.g.invdeg != nullptr ? da * p.g.invdeg[j] : da;
    }
  }
}

int spg_launch_ecc_step_bwd(const SpgEccStepBwd& p, hipStream_t stream) {
  const dim3 grid(spg_cdiv(p.g.N, 4));
  if (p.cell == SPG_CELL_LSTM) hipLaunchKernelGGL(spg_ecc_step_bwd_kernel<SPG_CELL_LSTM>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(spg_ecc_step_bwd_kernel<SPG_CELL_GRU>, grid, dim3(256), 0, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// per-edge filter gradient, summed over the R iterations in registers (written once)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spg_ecc_edge_wgrad_kernel(const SpgGraph g, int matrix,
                                                                 const float* __restrict__ states, long lds_,
                                                                 const float* __restrict__ G, long ldg, int R,
                                                                 float* __restrict__ dW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 4 + wave;
  if (e >= g.E) return;
  const float* hs = states + (long)g.src[e] * lds_;
  const float* gd = G + (long)g.dst[e] * ldg;
  if (matrix) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gd + r * 32 + 4 * (lane & 7));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float hk = hs[r * 32 + (lane >> 3) + 8 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(hk, g4[c], acc[q][c]);
      }
    }
    f32x4* o = reinterpret_cast<f32x4*>(dW + (long)e * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[lane + 64 * q] = acc[q];
  } else if (lane < 32) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a = fmaf(hs[r * 32 + lane], gd[r * 32 + lane], a);
    dW[(long)e * 32 + lane] = a;
  }
}

int spg_launch_ecc_edge_wgrad(const SpgGraph& g, int matrix, const float* states, long lds, const float* G, long ldg,
                              int R, float* dW, hipStream_t stream) {
  if (g.E == 0) return 0;
  hipLaunchKernelGGL(spg_ecc_edge_wgrad_kernel, dim3(spg_cdiv(g.E, 4)), dim3(256), 0, stream, g, matrix, states, lds, G,
                     ldg, R, dW);
  SPG_LAUNCH_CHECK();
  return 0;
}

__global__ void spg_copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long rows,
                                  int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  dst[r * ldd + c] = src[r * lds_ + c];
}

int spg_launch_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, hipStream_t stream) {
  const long n = rows * cols;
  if (n == 0) return 0;
  hipLaunchKernelGGL(spg_copy2d_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols);
  SPG_LAUNCH_CHECK();
  return 0;
}
