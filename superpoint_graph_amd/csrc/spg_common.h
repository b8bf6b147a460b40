// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the superpoint-graph hot path.
// Written for wave64 / MFMA f32 (v_mfma_f32_32x32x2_f32) only -- no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPG_KC 32          // reduction chunk (floats) staged in LDS per step
#define SPG_THREADS 256    // 4 wavefronts per workgroup, one per SIMD

// ----------------------------------------------------------------------------------------------
// error plumbing (C-ABI: every entry point returns 0 or a hipError_t / negative argument error)
// ----------------------------------------------------------------------------------------------
extern "C" const char* spg_last_error(void);
void spg_set_error(const char* fmt, ...);

#define SPG_CHECK_ARG(cond, msg)                                          \
  do {                                                                    \
    if (!(cond)) {                                                        \
      spg_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, msg);  \
      return -1;                                                          \
    }                                                                     \
  } while (0)

#define SPG_LAUNCH_CHECK()                                                                  \
  do {                                                                                      \
    hipError_t e__ = hipGetLastError();                                                     \
    if (e__ != hipSuccess) {                                                                \
      spg_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                                      \
    }                                                                                       \
  } while (0)

#define SPG_TRY(expr)           \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

static inline int spg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ----------------------------------------------------------------------------------------------
// Operand descriptor: a row-major matrix [rows, ld] (rows = points / superpoints / edges / nodes,
// columns = channels) read through a per-channel "prologue" so that BatchNorm+ReLU of the producer
// layer (forward) or the BatchNorm backward formula (backward) is fused into the LDS staging of
// the consumer GEMM and never materialised in HBM.
// ----------------------------------------------------------------------------------------------
enum {
  SPG_PRO_IDENT = 0,    // v = X[m, c]
  SPG_PRO_AFFINE = 1,   // v = X[m, c] * c0[c] + c1[c] (c < n_affine; c0 may be null), then ReLU if relu
  SPG_PRO_CLOUD = 2,    // X = clouds [G, Ctot, P] channel-major; m = g*P + p; optional 2x2 STN transform of channels 0,1
  SPG_PRO_BNBWD = 3,    // dz = X[m,c], y = X2[m,c]; v = c0[c]*(dz - c1[c]) - (y - c2[c])*c3[c]   (BatchNorm backward)
  SPG_PRO_POOLBWD = 4   // dz = (aidx[g,c] == p) ? X[g,c] : 0 (gradient of the max-pool), then as BNBWD
};

struct SpgOperand {
  const float* X;
  const float* X2;
  long ld;            // leading dimension of X (and X2) in floats
  const float* c0;
  const float* c1;
  const float* c2;
  const float* c3;
  const int* aidx;    // POOLBWD: [G, ldg] row index (within the group) that won the max-pool
  const float* stnT;  // CLOUD: raw STN projection output [G, 4] (identity is added here), or null
  int mode;
  int relu;
  int n_affine;       // AFFINE: channels >= n_affine pass through unchanged (concatenated global features)
  int Ctot;           // CLOUD: channels per cloud in memory
  int P;              // CLOUD / POOLBWD: rows (points) per group
  int ldg;            // POOLBWD: leading dimension of the per-group arrays X (=dpool) and aidx
};

__device__ __forceinline__ float spg_fetch(const SpgOperand& d, long m, int c) {
  switch (d.mode) {
    case SPG_PRO_IDENT:
      return d.X[m * d.ld + c];
    case SPG_PRO_AFFINE: {
      float v = d.X[m * d.ld + c];
      if (c < d.n_affine) {
        if (d.c0) v = fmaf(v, d.c0[c], d.c1[c]);
        if (d.relu) v = fmaxf(v, 0.f);
      }
      return v;
    }
    case SPG_PRO_CLOUD: {
      long g = m / d.P;
      int p = (int)(m - g * d.P);
      const float* base = d.X + (g * d.Ctot) * (long)d.P + p;
      if (d.stnT != nullptr && c < 2) {
        // learning/pointnet.py:123  [x y] @ (proj.view(2,2) + I)
        const float* T = d.stnT + g * 4;
        float x = base[0], y = base[d.P];
        return c == 0 ? fmaf(x, T[0] + 1.f, y * T[2]) : fmaf(x, T[1], y * (T[3] + 1.f));
      }
      return base[(long)c * d.P];
    }
    case SPG_PRO_BNBWD: {
      float dz = d.X[m * d.ld + c];
      float y = d.X2[m * d.ld + c];
      return d.c0[c] * (dz - d.c1[c]) - (y - d.c2[c]) * d.c3[c];
    }
    default: {  // SPG_PRO_POOLBWD
      long g = m / d.P;
      int p = (int)(m - g * d.P);
      float dz = (d.aidx[g * d.ldg + c] == p) ? d.X[g * d.ldg + c] : 0.f;
      float y = d.X2[m * d.ld + c];
      return d.c0[c] * (dz - d.c1[c]) - (y - d.c2[c]) * d.c3[c];
    }
  }
}

// 4 consecutive channels c..c+3 of one row; caller guarantees 16-byte alignment of the row segment
// and that the mode is IDENT / AFFINE / BNBWD.
__device__ __forceinline__ f32x4 spg_fetch4(const SpgOperand& d, long m, int c) {
  f32x4 v = *reinterpret_cast<const f32x4*>(d.X + m * d.ld + c);
  if (d.mode == SPG_PRO_AFFINE) {
    if (d.c0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c + i < d.n_affine) v[i] = fmaf(v[i], d.c0[c + i], d.c1[c + i]);
    }
    if (d.relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c + i < d.n_affine) v[i] = fmaxf(v[i], 0.f);
    }
  } else if (d.mode == SPG_PRO_BNBWD) {
    f32x4 y = *reinterpret_cast<const f32x4*>(d.X2 + m * d.ld + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = d.c0[c + i] * (v[i] - d.c1[c + i]) - (y[i] - d.c2[c + i]) * d.c3[c + i];
  }
  return v;
}

__host__ __device__ inline bool spg_operand_vec_ok(const SpgOperand& d) {
  if (!(d.mode == SPG_PRO_IDENT || d.mode == SPG_PRO_AFFINE || d.mode == SPG_PRO_BNBWD)) return false;
  if ((d.ld & 3) != 0 || (((uintptr_t)d.X) & 15) != 0) return false;
  if (d.mode == SPG_PRO_BNBWD && (((uintptr_t)d.X2) & 15) != 0) return false;
  return true;
}

// ----------------------------------------------------------------------------------------------
// LDS tile layout shared by every MFMA kernel here:  tile[q][row][4]  (q = reduction index / 4,
// `rows`+1 16-byte slots per plane).  A lane (row r = lane&31, half h = lane>>5) reads ONE
// ds_read_b128 per 8 reduction steps: plane 2g+h gives it 4 consecutive reduction indices, used as
// the A (or B) operand of 4 successive v_mfma_f32_32x32x2_f32 -- both operands use the same
// (plane, component) -> reduction-index map, so the pairing inside each MFMA is consistent.
// Consecutive rows are consecutive 16-byte slots: conflict-free for ds_read_b128 and ds_write_b128;
// the odd plane stride (rows+1) keeps the 8 planes of one row on distinct banks for the writes.
// ----------------------------------------------------------------------------------------------
template <int TI, int TJ>
__device__ __forceinline__ void spg_mfma_chunk(const f32x4* __restrict__ As, const f32x4* __restrict__ Bs,
                                               int strideA, int strideB, int rowA, int rowB, int h,
                                               f32x16 (&acc)[TI][TJ]) {
#pragma unroll
  for (int g = 0; g < SPG_KC / 8; ++g) {
    f32x4 a[TI], b[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) a[i] = As[(2 * g + h) * strideA + rowA + 32 * i];
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = Bs[(2 * g + h) * strideB + rowB + 32 * j];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
  }
}

// Stage a [nrows_tile x SPG_KC] tile, element (row, k) = operand(m0+row, k0+k), rows >= mvalid and
// channels >= nch are zero.  (forward / dgrad A operand: LDS row = matrix row, reduction = channel)
template <int ROWS>
__device__ __forceinline__ void spg_stage_rows(const SpgOperand& d, long m0, int mvalid, int k0, int nch,
                                               f32x4* __restrict__ lds, bool vec) {
  const int tid = threadIdx.x;
  if (vec) {
    // 8 float4 per row; a thread keeps the same k-quad for all its rows
    const int kq = tid & 7;
    const int c = k0 + 4 * kq;
#pragma unroll 4
    for (int row = tid >> 3; row < ROWS; row += SPG_THREADS / 8) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < mvalid && c < nch) v = spg_fetch4(d, m0 + row, c);   // nch % 4 == 0 on this path
      lds[kq * (ROWS + 1) + row] = v;
    }
  } else if (d.mode == SPG_PRO_CLOUD) {
    // channel-major source: consecutive lanes take consecutive points (coalesced along p)
    float* l = reinterpret_cast<float*>(lds);
    for (int idx = tid; idx < ROWS * SPG_KC; idx += SPG_THREADS) {
      const int row = idx % ROWS, k = idx / ROWS;
      float v = 0.f;
      if (row < mvalid && k0 + k < nch) v = spg_fetch(d, m0 + row, k0 + k);
      l[((k >> 2) * (ROWS + 1) + row) * 4 + (k & 3)] = v;
    }
  } else {
    float* l = reinterpret_cast<float*>(lds);
    for (int idx = tid; idx < ROWS * SPG_KC; idx += SPG_THREADS) {
      const int row = idx / SPG_KC, k = idx % SPG_KC;
      float v = 0.f;
      if (row < mvalid && k0 + k < nch) v = spg_fetch(d, m0 + row, k0 + k);
      l[((k >> 2) * (ROWS + 1) + row) * 4 + (k & 3)] = v;
    }
  }
}

// Stage a [CH x SPG_KC] tile of the TRANSPOSED operand: LDS row = channel (c0+ch), reduction index =
// matrix row (m0 + k).  Used by the weight-gradient kernel (reduction over points/edges/nodes).
// Each thread gathers 4 consecutive rows of one channel (4 coalesced 4-byte loads across the wave)
// and writes one 16-byte slot.
template <int CH>
__device__ __forceinline__ void spg_stage_cols(const SpgOperand& d, long m0, long mend, int c0, int nch,
                                               f32x4* __restrict__ lds) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < CH * (SPG_KC / 4); idx += SPG_THREADS) {
    const int ch = idx % CH, mq = idx / CH;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c0 + ch < nch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        long m = m0 + 4 * mq + i;
        if (m < mend) v[i] = spg_fetch(d, m, c0 + ch);
      }
    }
    lds[mq * (CH + 1) + ch] = v;
  }
}

// Weight tile: W [nout, kred] row-major (ld), LDS row = output channel n0+j, reduction = k0+k.
template <int JT>
__device__ __forceinline__ void spg_stage_weight(const float* __restrict__ W, long ld, int n0, int nout, int k0,
                                                 int kred, f32x4* __restrict__ lds, bool vec) {
  const int tid = threadIdx.x;
  if (vec) {
    const int kq = tid & 7;
    const int k = k0 + 4 * kq;
#pragma unroll 4
    for (int j = tid >> 3; j < JT; j += SPG_THREADS / 8) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + j < nout && k < kred) v = *reinterpret_cast<const f32x4*>(W + (long)(n0 + j) * ld + k);
      lds[kq * (JT + 1) + j] = v;
    }
  } else {
    float* l = reinterpret_cast<float*>(lds);
    for (int idx = tid; idx < JT * SPG_KC; idx += SPG_THREADS) {
      const int j = idx / SPG_KC, k = idx % SPG_KC;
      float v = 0.f;
      if (n0 + j < nout && k0 + k < kred) v = W[(long)(n0 + j) * ld + k0 + k];
      l[((k >> 2) * (JT + 1) + j) * 4 + (k & 3)] = v;
    }
  }
}

// C/D fragment of v_mfma_f32_32x32x2_f32: lane holds column (lane&31), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int spg_acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float spg_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
