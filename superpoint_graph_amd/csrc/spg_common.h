// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the superpoint-graph hot path.
// Written for wave64 / MFMA f32 (v_mfma_f32_32x32x2_f32) only -- no other target is supported.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPG_KC 32          // reduction chunk (floats) staged in LDS per step
#define SPG_MAX_DEVICES 16
#define SPG_THREADS 256    // 4 wavefronts per workgroup, one per SIMD

// ----------------------------------------------------------------------------------------------
// error plumbing (C-ABI: every entry point returns 0 or a hipError_t / negative argument error)
// ----------------------------------------------------------------------------------------------
extern "C" const char* spg_last_error(void);
void spg_set_error(const char* fmt, ...);
// tuning knobs (process-global; defaults are the production values): see spg_tune in include/spg_hip.h
enum { SPG_TUNE_NO_PERSIST = 0, SPG_TUNE_DBG = 3, SPG_TUNE_FIN_SLICE_MIN = 4, SPG_TUNE_NO_STAT_ACCUM = 5, SPG_TUNE_NO_HEAD_SERVICE = 6, SPG_TUNE_PRECISION = 7,
       SPG_TUNE_NO_PERSIST_ECC = 8, SPG_TUNE_SIDE_STREAM = 9, SPG_TUNE_NO_BN_FOLD = 10, SPG_TUNE_NO_GROUP = 11, SPG_TUNE_SPLITK = 12, SPG_TUNE_NO_VEC_GENERIC = 13, SPG_TUNE_NO_BWD_PAIR = 14, SPG_TUNE_NO_ECC_HEAD = 15, SPG_TUNE_LEAVES = 16, SPG_TUNE_NO_NARROW_PAIR = 17, SPG_TUNE_NO_FIRST_CONV_BWD = 18, SPG_TUNE_NO_OWNER_FIRST = 19, SPG_TUNE_PX_SPIN_LIMIT = 20, SPG_TUNE_NO_ADAM_GUARD = 21, SPG_TUNE_NO_PAIR_SPLIT = 22, SPG_TUNE_COUNT = 23 };
int spg_tune_get(int key);

// Fork / join of a library-owned side stream (one per device, created on first use): a latency-bound chain of small-grid
// launches that does not depend on the caller's next launches runs NEXT to them instead of in front of them.  Every launch of
// this step costs >= ~4.5 us of fixed latency (dispatch, first loads, end-of-kernel release) whatever it computes, and the
// small grids leave most CUs idle, so two independent small chains genuinely overlap.  fork: the side stream waits for
// everything enqueued on `main` so far; join: `main` waits for everything enqueued on the side stream.  Both asynchronous; a
// call that forks always joins before it returns, so the caller's stream semantics are unchanged.  Returns null (and the caller
// stays on `main`) unless the facility is switched ON (spg_tune key 9 = 1) and a stream can be created.
// MEASURED (round 3, profiles/r03_side_stream_experiment.txt): with the recurrent cell's five parameter-gradient launches
// (43 us) forked next to the filter network's backward chain the step got 50 us SLOWER (1.634 -> 1.684 ms, same box,
// interleaved runs): the two cross-stream dependencies cost more than the overlapped work is worth.  Hence off by default.
hipStream_t spg_side_fork(hipStream_t main);
int spg_side_join(hipStream_t main);

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

// BatchNorm-backward prologue value a * (x - b) - (y - c) * d with ONE rounding sequence everywhere: the plain expression
// leaves the compiler two ways to contract it into a fused multiply-add (either product can become the fma), and its choice
// depends on the surrounding code -- the same body inlined into two kernels (stand-alone launch / grouped launch, scalar /
// vector staging) then differs in the last bit.  sub, sub, mul (never contracted), fma: four VALU operations as before.
__device__ __forceinline__ float spg_bnbwd_value(float a, float x, float b, float y, float c, float d) {
  return fmaf(a, x - b, -__fmul_rn(y - c, d));
}

// PACKED forms (round 6): gfx950 has two-wide fp32 VALU operations (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 on aligned
// register pairs) and an fp32 MFMA shares the SIMD's issue with EVERY VALU operation -- a packed one costs what a plain one costs
// (tools/probe/coissue_probe.hip: 8 v_fma_f32 or 8 v_pk_fma_f32 behind each MFMA: 108.2 cycles per MFMA either way), so the staging
// and epilogue arithmetic of the fp32 GEMM kernels is written on float2 / float4 values where the elements are independent.
// Element by element these are the SAME IEEE operations in the same order as the scalar forms above (bit-identical results).
typedef float spg_f32x2 __attribute__((ext_vector_type(2)));
typedef float spg_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ spg_f32x4 spg_fma4(spg_f32x4 a, spg_f32x4 b, spg_f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ spg_f32x2 spg_fma2(spg_f32x2 a, spg_f32x2 b, spg_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ spg_f32x4 spg_bnbwd_value4(spg_f32x4 a, spg_f32x4 x, spg_f32x4 b, spg_f32x4 y, spg_f32x4 c, spg_f32x4 d) {
#pragma clang fp contract(off)
  const spg_f32x4 t = (y - c) * d;          // (never contracted: see spg_bnbwd_value)
  const spg_f32x4 u = x - b;
  return __builtin_elementwise_fma(a, u, -t);
}

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
      return spg_bnbwd_value(d.c0[c], dz, d.c1[c], y, d.c2[c], d.c3[c]);
    }
    default: {  // SPG_PRO_POOLBWD
      long g = m / d.P;
      int p = (int)(m - g * d.P);
      float dz = (d.aidx[g * d.ldg + c] == p) ? d.X[g * d.ldg + c] : 0.f;
      float y = d.X2[m * d.ld + c];
      return spg_bnbwd_value(d.c0[c], dz, d.c1[c], y, d.c2[c], d.c3[c]);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Vectorised fetch: 4 consecutive channels of one row with the per-channel constants hoisted into registers
// (SpgQuad): loaded once per K-chunk (forward / dgrad) or once per kernel (weight gradient, where a thread keeps
// the same channel quad for all rows).  The operand mode is a TEMPLATE parameter on this path: the main loops
// must be straight-line code (no per-item branches) so that the compiler can keep the global loads of the
// next chunk in flight behind the MFMAs of the current one with counted s_waitcnt.
// ----------------------------------------------------------------------------------------------
struct SpgQuad {
  f32x4 a, b, c, d;
  int nvalid;    // number of valid channels in this quad (0..4); the others are forced to zero
  int naff;      // AFFINE: number of leading channels of the quad that get scale/shift(/ReLU); the rest pass through
  int relu;
  int has;       // AFFINE: scale / shift arrays present (otherwise ReLU only)
};

__host__ __device__ inline bool spg_operand_vec_ok(const SpgOperand& d) {
  if (d.mode == SPG_PRO_CLOUD) return false;
  if ((d.ld & 3) != 0 || (((uintptr_t)d.X) & 15) != 0) return false;
  if ((d.mode == SPG_PRO_BNBWD || d.mode == SPG_PRO_POOLBWD) && (((uintptr_t)d.X2) & 15) != 0) return false;
  if (d.mode == SPG_PRO_POOLBWD && ((d.ldg & 3) != 0 || (((uintptr_t)d.aidx) & 15) != 0)) return false;
  // per-channel constants are fetched as aligned quads
  if (d.mode == SPG_PRO_AFFINE &&
      ((d.n_affine & 3) != 0 || (d.c0 != nullptr && ((((uintptr_t)d.c0) | ((uintptr_t)d.c1)) & 15) != 0))) return false;
  if ((d.mode == SPG_PRO_BNBWD || d.mode == SPG_PRO_POOLBWD) &&
      ((((uintptr_t)d.c0) | ((uintptr_t)d.c1) | ((uintptr_t)d.c2) | ((uintptr_t)d.c3)) & 15) != 0) return false;
  return true;
}

// constants of channels c..c+3 (nch = number of channels of the operand).  All loads are unconditional (clamped
// indices) and NOT consumed here, so they travel with the data loads of the chunk; masking happens in
// spg_finish_raw.
template <int MODE>
__device__ __forceinline__ SpgQuad spg_quad_consts(const SpgOperand& d, int c, int nch) {
  SpgQuad q;
  q.nvalid = nch - c < 0 ? 0 : (nch - c > 4 ? 4 : nch - c);
  q.naff = 0; q.relu = 0; q.has = 0;
  if (MODE == SPG_PRO_AFFINE) {
    // n_affine is a multiple of 4 (spg_operand_vec_ok): a quad is either fully affine or passes through
    const int lim = d.n_affine < nch ? d.n_affine : nch;
    q.naff = c + 4 <= lim ? 4 : 0;
    q.relu = d.relu;
    q.has = d.c0 != nullptr;
    // no branch: without scale / shift the loads read the (always readable) operand itself and are ignored
    const float* p0 = q.has ? d.c0 : d.X;
    const float* p1 = q.has ? d.c1 : d.X;
    const int cb = (q.has && q.naff) ? c : 0;
    q.a = *reinterpret_cast<const f32x4*>(p0 + cb);
    q.b = *reinterpret_cast<const f32x4*>(p1 + cb);
  } else if (MODE == SPG_PRO_BNBWD || MODE == SPG_PRO_POOLBWD) {
    // a partial last quad reads up to 3 floats past nch inside the [4][C] constant block; its lanes are masked
    const int cb = c < nch ? c : 0;
    q.a = *reinterpret_cast<const f32x4*>(d.c0 + cb);
    q.b = *reinterpret_cast<const f32x4*>(d.c1 + cb);
    q.c = *reinterpret_cast<const f32x4*>(d.c2 + cb);
    q.d = *reinterpret_cast<const f32x4*>(d.c3 + cb);
  } else if (MODE == SPG_PRO_CLOUD) {
    q.naff = (d.stnT != nullptr && c == 0) ? 1 : 0;         // this quad holds x, y: apply the 2x2 STN transform
  }
  return q;
}

// Raw (un-processed) global data of one quad: the loads are issued early (software pipelining: they are in flight
// while the MFMAs of the previous chunk run) and finished -- prologue arithmetic -- right before the LDS write.
struct SpgRaw {
  f32x4 x, y;
  int4 ai;     // POOLBWD: arg-max point of the (group, channel)s of the quad
  int pp;      // POOLBWD: point index of this row inside its group
};

// rows m[i] (already clamped into the matrix), channels c..c+3 (clamped; each row segment must be readable as one
// aligned 16-byte load: ld % 4 == 0).  All loads are unconditional.  Load order: primary loads (and, for the
// max-pool backward, the tiny per-group arg-max rows), then the secondary loads.  NOTHING loaded here is consumed
// here: every use (masks included) sits in spg_finish_raw, behind the MFMAs of the previous chunk.
template <int MODE, int NI>
__device__ __forceinline__ void spg_load_raw(const SpgOperand& d, const long (&m)[NI], int c, int nvalid, SpgRaw (&r)[NI]) {
  if (MODE == SPG_PRO_CLOUD) {
    // channel-major clouds [G, Ctot, P]: four dword loads per quad (the caller maps lanes to consecutive points)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned mu = (unsigned)m[i], P = (unsigned)d.P;
      const unsigned g = mu / P;
      const float* base = d.X + ((long)g * d.Ctot) * d.P + (mu - g * P);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[i].x[e] = base[(long)(e < nvalid ? c + e : 0) * d.P];
      if (d.stnT != nullptr) r[i].y = *reinterpret_cast<const f32x4*>(d.stnT + (long)g * 4);   // wave-uniform branch
    }
  } else if (MODE == SPG_PRO_POOLBWD) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned mu = (unsigned)m[i], P = (unsigned)d.P;
      const unsigned g = mu / P;
      r[i].pp = (int)(mu - g * P);
      r[i].x = *reinterpret_cast<const f32x4*>(d.X + (long)g * d.ldg + c);
      r[i].ai = *reinterpret_cast<const int4*>(d.aidx + (long)g * d.ldg + c);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i].y = *reinterpret_cast<const f32x4*>(d.X2 + m[i] * d.ld + c);
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i].x = *reinterpret_cast<const f32x4*>(d.X + m[i] * d.ld + c);
    if (MODE == SPG_PRO_BNBWD) {
#pragma unroll
      for (int i = 0; i < NI; ++i) r[i].y = *reinterpret_cast<const f32x4*>(d.X2 + m[i] * d.ld + c);
    }
  }
}

// one row quad (same contract as spg_load_raw)
template <int MODE>
__device__ __forceinline__ void spg_load_raw1(const SpgOperand& d, long m, int c, int nvalid, SpgRaw& r) {
  const long mm[1] = {m};
  SpgRaw rr[1];
  spg_load_raw<MODE, 1>(d, mm, c, nvalid, rr);
  r = rr[0];
}

template <int MODE>
__device__ __forceinline__ f32x4 spg_finish_raw(const SpgQuad& q, const SpgRaw& r, bool valid) {
  f32x4 v;
  if (MODE == SPG_PRO_IDENT) {
    v = r.x;
  } else if (MODE == SPG_PRO_CLOUD) {
    v = r.x;
    if (q.naff) {   // learning/pointnet.py:123  [x y] @ (proj.view(2,2) + I); same expression as spg_fetch
      const float x = r.x[0], y = r.x[1];
      v[0] = fmaf(x, r.y[0] + 1.f, y * r.y[2]);
      v[1] = fmaf(x, r.y[1], y * (r.y[3] + 1.f));
    }
  } else if (MODE == SPG_PRO_AFFINE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = q.has ? fmaf(r.x[i], q.a[i], q.b[i]) : r.x[i];
      v[i] = q.naff ? (q.relu ? fmaxf(t, 0.f) : t) : r.x[i];
    }
  } else if (MODE == SPG_PRO_POOLBWD) {
    // the gradient of the max-pool goes to the winning row of each (group, channel)
    const float g0 = r.ai.x == r.pp ? r.x[0] : 0.f, g1 = r.ai.y == r.pp ? r.x[1] : 0.f;
    const float g2 = r.ai.z == r.pp ? r.x[2] : 0.f, g3 = r.ai.w == r.pp ? r.x[3] : 0.f;
    v[0] = spg_bnbwd_value(q.a[0], g0, q.b[0], r.y[0], q.c[0], q.d[0]);
    v[1] = spg_bnbwd_value(q.a[1], g1, q.b[1], r.y[1], q.c[1], q.d[1]);
    v[2] = spg_bnbwd_value(q.a[2], g2, q.b[2], r.y[2], q.c[2], q.d[2]);
    v[3] = spg_bnbwd_value(q.a[3], g3, q.b[3], r.y[3], q.c[3], q.d[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = spg_bnbwd_value(q.a[i], r.x[i], q.b[i], r.y[i], q.c[i], q.d[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (valid && i < q.nvalid) ? v[i] : 0.f;
  return v;
}

// ----------------------------------------------------------------------------------------------
// LDS tile layouts.
//  "out-major"  tile[q][row][4]  (q = reduction index / 4, rows+1 16-byte slots per plane): natural when the
//     global matrix is [row][reduction] (forward / dgrad A operand, forward weights).  A lane (row r = lane&31,
//     half h = lane>>5) reads ONE ds_read_b128 per 8 reduction steps: plane 2g+h holds reduction indices
//     8g+4h+{0..3}, used as the operand of 4 successive v_mfma_f32_32x32x2_f32.  Conflict-free b128 reads and
//     writes (consecutive rows = consecutive slots; odd plane stride).
//  "red-major"  tile[red][ch] row-major floats (stride = channels + pad): natural when the global matrix is
//     [reduction][channel] (both operands of the weight gradient, the untransposed weights of the dgrad).
//     Written with ds_write_b128 exactly as loaded (float4 along the channels), read with one ds_read_b32 per
//     MFMA operand (32 consecutive channels = 32 consecutive banks).  LDS bandwidth is irrelevant here: a
//     wave issues TI+TJ dword reads per TI*TJ MFMAs of 64 cycles each.
// Both layouts use the same reduction-index <-> (MFMA k-slot, lane half) map, so they can be mixed.
// ----------------------------------------------------------------------------------------------
template <int TI, int TJ>
__device__ __forceinline__ void spg_mfma_chunk(const f32x4* __restrict__ As, const f32x4* __restrict__ Bs,
                                               int strideA, int strideB, int rowA, int rowB, int h,
                                               f32x16 (&acc)[TI][TJ]) {
  // register double-buffering of the LDS fragments: the reads of group g+1 are issued before the MFMAs of group g
  f32x4 a[2][TI], b[2][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) a[0][i] = As[h * strideA + rowA + 32 * i];
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[0][j] = Bs[h * strideB + rowB + 32 * j];
#pragma unroll
  for (int g = 0; g < SPG_KC / 8; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    if (g + 1 < SPG_KC / 8) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[nxt][i] = As[(2 * (g + 1) + h) * strideA + rowA + 32 * i];
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[nxt][j] = Bs[(2 * (g + 1) + h) * strideB + rowB + 32 * j];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][s], b[cur][j][s], acc[i][j], 0, 0, 0);
  }
}

// ---- software-pipelined variants --------------------------------------------------------------------------
// A wave issues in order: while it sits in a run of back-to-back MFMAs (64 cycles each) it can do nothing else, and
// while it stages the next chunk its MFMA pipe idles.  The *_il variants take a functor `piece(slot)` and call it after
// every MFMA k-step (16 slots per chunk), fenced by scheduling barriers, so the staging work of the NEXT chunks (global
// loads, prologue arithmetic, LDS writes) is issued in the shadow of the matrix pipe instead of between two chunks.
template <int TI, int TJ, class Piece>
__device__ __forceinline__ void spg_mfma_chunk_il(const f32x4* __restrict__ As, const f32x4* __restrict__ Bs,
                                                  int strideA, int strideB, int rowA, int rowB, int h,
                                                  f32x16 (&acc)[TI][TJ], Piece&& piece) {
  f32x4 a[2][TI], b[2][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) a[0][i] = As[h * strideA + rowA + 32 * i];
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[0][j] = Bs[h * strideB + rowB + 32 * j];
#pragma unroll
  for (int g = 0; g < SPG_KC / 8; ++g) {
    const int cur = g & 1, nxt = cur ^ 1;
    if (g + 1 < SPG_KC / 8) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[nxt][i] = As[(2 * (g + 1) + h) * strideA + rowA + 32 * i];
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[nxt][j] = Bs[(2 * (g + 1) + h) * strideB + rowB + 32 * j];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][s], b[cur][j][s], acc[i][j], 0, 0, 0);
      piece(4 * g + s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int TI, int TJ, class Piece>
__device__ __forceinline__ void spg_mfma_chunk_or_il(const f32x4* __restrict__ As, const float* __restrict__ Bs,
                                                     int strideA, int strideB, int rowA, int colB, int h,
                                                     f32x16 (&acc)[TI][TJ], Piece&& piece) {
  f32x4 a[2][TI];
  float b[2][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) a[0][i] = As[h * strideA + rowA + 32 * i];
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[0][j] = Bs[(4 * h) * strideB + colB + 32 * j];
#pragma unroll
  for (int g = 0; g < SPG_KC / 8; ++g) {
    if (g + 1 < SPG_KC / 8) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[(g + 1) & 1][i] = As[(2 * (g + 1) + h) * strideA + rowA + 32 * i];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t = 4 * g + s;          // k-step inside the chunk
      if (t + 1 < SPG_KC / 2) {
        const int g1 = (t + 1) >> 2, s1 = (t + 1) & 3;
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[(t + 1) & 1][j] = Bs[(8 * g1 + 4 * h + s1) * strideB + colB + 32 * j];
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][s], b[t & 1][j], acc[i][j], 0, 0, 0);
      piece(t);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int TI, int TJ, class Piece>
__device__ __forceinline__ void spg_mfma_chunk_rr_il(const float* __restrict__ As, const float* __restrict__ Bs,
                                                     int strideA, int strideB, int colA, int colB, int h,
                                                     f32x16 (&acc)[TI][TJ], Piece&& piece) {
  float a[2][TI], b[2][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) a[0][i] = As[h * strideA + colA + 32 * i];
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[0][j] = Bs[h * strideB + colB + 32 * j];
#pragma unroll
  for (int kk = 0; kk < SPG_KC / 2; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < SPG_KC / 2) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[nxt][i] = As[(2 * (kk + 1) + h) * strideA + colA + 32 * i];
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[nxt][j] = Bs[(2 * (kk + 1) + h) * strideB + colB + 32 * j];
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    piece(kk);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A out-major, B red-major
template <int TI, int TJ>
__device__ __forceinline__ void spg_mfma_chunk_or(const f32x4* __restrict__ As, const float* __restrict__ Bs,
                                                  int strideA, int strideB, int rowA, int colB, int h,
                                                  f32x16 (&acc)[TI][TJ]) {
  // register double-buffering of the LDS fragments, as in spg_mfma_chunk: the reads of group g+1 are issued before the MFMAs of
  // group g (left to the compiler the first MFMA of every group waits for the reads issued right in front of it: one LDS
  // round trip per 8 MFMAs of a wave with a single accumulator chain -- the data-gradient waves of spg_bwdpair_kernel)
  f32x4 a[2][TI];
  float b[2][4][TJ];
  auto fetch = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TI; ++i) a[buf][i] = As[(2 * g + h) * strideA + rowA + 32 * i];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[buf][s][j] = Bs[(8 * g + 4 * h + s) * strideB + colB + 32 * j];
  };
  fetch(0, 0);
#pragma unroll
  for (int g = 0; g < SPG_KC / 8; ++g) {
    const int cur = g & 1;
    if (g + 1 < SPG_KC / 8) fetch(cur ^ 1, g + 1);
    __builtin_amdgcn_sched_barrier(0);      // (the next group's reads stay in front of this group's MFMAs ...)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][s], b[cur][s][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);      // (... and this group's MFMAs in front of the reads after next)
  }
}

// A out-major, B red-major, ONE 32 x 32 output block, over NG groups of 8 reduction indices (NG / 4 chunks of one LDS tile: the plane
// and row formulas continue across chunk boundaries) with the fragment reads DEPTH groups ahead of their MFMAs.  A wave with one
// accumulator chain (the data-gradient waves of spg_bwdpair_kernel) stalls on every exposed LDS round trip, and a chunk-by-chunk loop
// exposes one per chunk; measured (round 6, profiles/r06_bwdpair_roles.txt): its 64 MFMAs of a tile took 9300 cycles next to the
// weight-gradient wave's, 4400 with this loop.  Same MFMA order as NG / 4 calls of spg_mfma_chunk_or: bit-identical.
template <int NG, int DEPTH>
__device__ __forceinline__ void spg_mfma_tile_or(const f32x4* __restrict__ As, const float* __restrict__ Bs,
                                                 int strideA, int strideB, int rowA, int colB, int h, f32x16& acc) {
  constexpr int RING = DEPTH + 1;
  f32x4 a[RING];
  float b[RING][4];
  auto fetch = [&](int slot, int g) __attribute__((always_inline)) {
    a[slot] = As[(2 * g + h) * strideA + rowA];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[slot][s] = Bs[(8 * g + 4 * h + s) * strideB + colB];
  };
#pragma unroll
  for (int d = 0; d < DEPTH && d < NG; ++d) fetch(d, d);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + DEPTH < NG) fetch((g + DEPTH) % RING, g + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g % RING][s], b[g % RING][s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// both operands red-major
template <int TI, int TJ>
__device__ __forceinline__ void spg_mfma_chunk_rr(const float* __restrict__ As, const float* __restrict__ Bs,
                                                  int strideA, int strideB, int colA, int colB, int h,
                                                  f32x16 (&acc)[TI][TJ]) {
#pragma unroll
  for (int kk = 0; kk < SPG_KC / 2; ++kk) {
    float a[TI], b[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) a[i] = As[(2 * kk + h) * strideA + colA + 32 * i];
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = Bs[(2 * kk + h) * strideB + colB + 32 * j];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// A = TRANSPOSE of an out-major tile (reduction over its ROWS, its channels become the rows of the product), B red-major:
// the weight gradient computed from the very LDS copy of dz the data gradient reads as its out-major A operand
// (spg_bwdpair_kernel).  As: floats of the whole out-major tile [channel quad][strideA4 slots][4]; row0: first row of the chunk.
// Element (row, ch) sits at ((ch / 4) * strideA4 + row) * 4 + ch % 4: 32 consecutive channels = 32 different banks.
template <int TI, int TJ, int TA>
__device__ __forceinline__ void spg_mfma_chunk_tr(const float* __restrict__ As, const float* __restrict__ Bs,
                                                  int strideA4, int strideB, int row0, int colA, int colB, int h,
                                                  f32x16 (&acc)[TA][TJ]) {      // the first TI rows of acc are used
  static_assert(TI <= TA, "accumulator array too small");
  int oa[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = colA + 32 * i;
    oa[i] = ((co >> 2) * strideA4 + row0 + h) * 4 + (co & 3);
  }
  const float* __restrict__ Bh = Bs + h * strideB + colB;
#pragma unroll
  for (int kk = 0; kk < SPG_KC / 2; ++kk) {
    float a[TI], b[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) a[i] = As[oa[i] + 8 * kk];
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[j] = Bh[2 * kk * strideB + 32 * j];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// the same with the B operand finished on the fly: b = ReLU(raw * sc + sh), sc / sh of the lane's (fixed) channel.
// The LDS operand reads run DEPTH k-steps ahead of the MFMAs that use them (register ring, pinned by scheduling barriers): left
// to the compiler every k-step was `ds_read ...; s_waitcnt lgkmcnt; v_mfma` -- one exposed LDS round trip per TI MFMAs of the
// weight-gradient waves of spg_bwdpair_kernel.  Same MFMA order, same values: bit-identical.
// NK k-steps = 2 NK rows of the tile from row0 on (NK = SPG_KC / 2: one chunk; more: the row formulas continue across chunks).
template <int TI, int TA, int NK = SPG_KC / 2, int DEPTH = (TI >= 4 ? 1 : 2)>
__device__ __forceinline__ void spg_mfma_chunk_tr_aff(const float* __restrict__ As, const float* __restrict__ Bs,
                                                      int strideA4, int strideB, int row0, int colA, int colB, int h,
                                                      float sc, float sh, f32x16 (&acc)[TA][1]) {
  static_assert(TI <= TA, "accumulator array too small");
  constexpr int RING = DEPTH + 1;
  int oa[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = colA + 32 * i;
    oa[i] = ((co >> 2) * strideA4 + row0 + h) * 4 + (co & 3);
  }
  const float* __restrict__ Bh = Bs + h * strideB + colB;
  float a[RING][TI], braw[RING];
  auto fetch = [&](int slot, int kk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TI; ++i) a[slot][i] = As[oa[i] + 8 * kk];
    braw[slot] = Bh[2 * kk * strideB];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, d);
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    if (kk + DEPTH < NK) fetch((kk + DEPTH) % RING, kk + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
    const float b = fmaxf(fmaf(braw[kk % RING], sc, sh), 0.f);
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk % RING][i], b, acc[i][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Generic (runtime-mode, scalar) out-major staging of a [ROWS x SPG_KC] tile: element (row, k) =
// operand(m0+row, k0+k); rows >= mvalid and channels >= nch are zero.  Used when the vector path does not apply
// (channel-major clouds, unaligned leading dimensions); those GEMMs are tiny.
template <int ROWS>
__device__ __forceinline__ void spg_stage_rows(const SpgOperand& d, long m0, int mvalid, int k0, int nch,
                                               f32x4* __restrict__ lds) {
  const int tid = threadIdx.x;
  float* l = reinterpret_cast<float*>(lds);
  if (d.mode == SPG_PRO_CLOUD) {
    // channel-major source: consecutive lanes take consecutive points (coalesced along p)
    for (int idx = tid; idx < ROWS * SPG_KC; idx += SPG_THREADS) {
      const int row = idx % ROWS, k = idx / ROWS;
      float v = 0.f;
      if (row < mvalid && k0 + k < nch) v = spg_fetch(d, m0 + row, k0 + k);
      l[((k >> 2) * (ROWS + 1) + row) * 4 + (k & 3)] = v;
    }
  } else {
    for (int idx = tid; idx < ROWS * SPG_KC; idx += SPG_THREADS) {
      const int row = idx / SPG_KC, k = idx % SPG_KC;
      float v = 0.f;
      if (row < mvalid && k0 + k < nch) v = spg_fetch(d, m0 + row, k0 + k);
      l[((k >> 2) * (ROWS + 1) + row) * 4 + (k & 3)] = v;
    }
  }
}

// Generic (scalar) red-major staging of a [SPG_KC rows x CH channels] tile: LDS[r][ch] = operand(m0 + r, c0 + ch).
// `stride` (floats): CH+4 normally (16-byte aligned rows for the vector path), CH+1 for the channel-major cloud
// source (conflict-free dword writes with the points along the lanes).
template <int CH>
__device__ __forceinline__ int spg_red_stride(const SpgOperand& d) { return d.mode == SPG_PRO_CLOUD ? CH + 1 : CH + 4; }

template <int CH>
__device__ __forceinline__ void spg_stage_red(const SpgOperand& d, long m0, long mend, int c0, int nch,
                                              float* __restrict__ lds) {
  const int tid = threadIdx.x;
  if (d.mode == SPG_PRO_CLOUD) {
    for (int idx = tid; idx < CH * SPG_KC; idx += SPG_THREADS) {
      const int r = idx % SPG_KC, ch = idx / SPG_KC;
      float v = 0.f;
      if (m0 + r < mend && c0 + ch < nch) v = spg_fetch(d, m0 + r, c0 + ch);
      lds[r * (CH + 1) + ch] = v;
    }
  } else {
    for (int idx = tid; idx < CH * SPG_KC; idx += SPG_THREADS) {
      const int ch = idx % CH, r = idx / CH;
      float v = 0.f;
      if (m0 + r < mend && c0 + ch < nch) v = spg_fetch(d, m0 + r, c0 + ch);
      lds[r * (CH + 4) + ch] = v;
    }
  }
}

// Generic (scalar) out-major weight tile: W [nout, kred] row-major (ld), LDS row = output channel n0+j.
template <int JT>
__device__ __forceinline__ void spg_stage_weight(const float* __restrict__ W, long ld, int n0, int nout, int k0,
                                                 int kred, f32x4* __restrict__ lds) {
  float* l = reinterpret_cast<float*>(lds);
  for (int idx = threadIdx.x; idx < JT * SPG_KC; idx += SPG_THREADS) {
    const int j = idx / SPG_KC, k = idx % SPG_KC;
    float v = 0.f;
    if (n0 + j < nout && k0 + k < kred) v = W[(long)(n0 + j) * ld + k0 + k];
    l[((k >> 2) * (JT + 1) + j) * 4 + (k & 3)] = v;
  }
}

// Generic (scalar) red-major weight tile for the data gradient: W [kred, nout] row-major (ld) read UNTRANSPOSED:
// LDS[r][j] = W[k0 + r][n0 + j]   (reduction over the rows of W = output channels of the forward layer)
template <int JT>
__device__ __forceinline__ void spg_stage_weight_red(const float* __restrict__ W, long ld, int n0, int nout, int k0,
                                                     int kred, float* __restrict__ lds) {
  for (int idx = threadIdx.x; idx < JT * SPG_KC; idx += SPG_THREADS) {
    const int j = idx % JT, r = idx / JT;
    float v = 0.f;
    if (k0 + r < kred && n0 + j < nout) v = W[(long)(k0 + r) * ld + n0 + j];
    lds[r * (JT + 4) + j] = v;
  }
}

// ----------------------------------------------------------------------------------------------
// Software-pipelined staging (vector paths only): `load` issues the global loads of a chunk into registers,
// `store` finishes them (prologue arithmetic) and writes the LDS tile.  The kernels call load(c+1) before the
// MFMAs of chunk c and store(c+1) after them, into the other LDS buffer: HBM/L2 latency hides under the MFMAs.
// ----------------------------------------------------------------------------------------------
template <int MODE, int ROWS>
struct SpgRowsPipe {          // out-major [ROWS x 32] tile of an operand
  static constexpr int NI = ROWS / 32;
  SpgRaw raw[NI];
  SpgQuad q;
  unsigned vmask;             // bit i: row i of this thread is inside the tile
  __device__ __forceinline__ void load(const SpgOperand& d, long m0, int mvalid, int k0, int nch) {
    const int tid = threadIdx.x, c = k0 + 4 * (tid & 7);
    q = spg_quad_consts<MODE>(d, c, nch);
    long m[NI];
    vmask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const bool ok = row < mvalid;
      m[i] = m0 + (ok ? row : 0);                           // clamped: loads are unconditional
      vmask |= ok ? (1u << i) : 0u;
    }
    spg_load_raw<MODE, NI>(d, m, q.nvalid > 0 ? c : 0, q.nvalid, raw);
  }
  __device__ __forceinline__ void store(f32x4* __restrict__ lds) const {
    const int tid = threadIdx.x, kq = tid & 7;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      lds[kq * (ROWS + 1) + (tid >> 3) + 32 * i] = spg_finish_raw<MODE>(q, raw[i], (vmask >> i) & 1u);
  }
  // the same in parts, for the interleaved main loop: prepare(k0) once per chunk, then load_part(i) for every i;
  // later store_part(i) for every i
  int cq;
  __device__ __forceinline__ void prepare(const SpgOperand& d, int k0, int nch) {
    cq = k0 + 4 * (threadIdx.x & 7);
    q = spg_quad_consts<MODE>(d, cq, nch);
    vmask = 0;
  }
  __device__ __forceinline__ void load_part(const SpgOperand& d, long m0, int mvalid, int i) {
    const int row = (threadIdx.x >> 3) + 32 * i;
    const bool ok = row < mvalid;
    vmask |= ok ? (1u << i) : 0u;
    spg_load_raw1<MODE>(d, m0 + (ok ? row : 0), q.nvalid > 0 ? cq : 0, q.nvalid, raw[i]);
  }
  __device__ __forceinline__ void store_part(f32x4* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    lds[(tid & 7) * (ROWS + 1) + (tid >> 3) + 32 * i] = spg_finish_raw<MODE>(q, raw[i], (vmask >> i) & 1u);
  }
};

template <int MODE, int CH>
struct SpgRedPipe {           // red-major [32 x CH] tile of an operand (the thread's channel quad is fixed: q set once)
  static constexpr int QUADS = CH / 4, RPP = SPG_THREADS / QUADS, NI = SPG_KC / RPP;
  SpgRaw raw[NI];
  unsigned vmask;
  // row-major sources: consecutive lanes take consecutive channel quads of one row (16-byte loads, coalesced along
  // the channels); channel-major clouds: consecutive lanes take consecutive points of one channel quad.
  static __device__ __forceinline__ int quad_of(int tid) { return MODE == SPG_PRO_CLOUD ? tid / SPG_KC : tid % QUADS; }
  static __device__ __forceinline__ int row_of(int tid, int i) {
    return MODE == SPG_PRO_CLOUD ? (tid % SPG_KC) : tid / QUADS + RPP * i;
  }
  static constexpr int NITEMS = MODE == SPG_PRO_CLOUD ? (QUADS * SPG_KC + SPG_THREADS - 1) / SPG_THREADS : NI;
  __device__ __forceinline__ void load(const SpgOperand& d, const SpgQuad& q, long m0, long mend, int c0) {
    const int tid = threadIdx.x, c = c0 + 4 * quad_of(tid);
    long m[NI];
    vmask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const long mm = m0 + row_of(tid, i);
      const bool ok = mm < mend && (MODE != SPG_PRO_CLOUD || (i == 0 && quad_of(tid) < QUADS));
      m[i] = ok ? mm : m0;
      vmask |= ok ? (1u << i) : 0u;
    }
    spg_load_raw<MODE, NI>(d, m, q.nvalid > 0 ? c : 0, q.nvalid, raw);
  }
  __device__ __forceinline__ void store(const SpgQuad& q, float* __restrict__ lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (MODE == SPG_PRO_CLOUD && (i > 0 || quad_of(tid) >= QUADS)) continue;   // 32 rows x QUADS quads <= 256 threads
      *reinterpret_cast<f32x4*>(lds + row_of(tid, i) * (CH + 4) + 4 * quad_of(tid)) =
          spg_finish_raw<MODE>(q, raw[i], (vmask >> i) & 1u);
    }
  }
  // part-wise form for the interleaved main loop: rows outside [mchunk, mend) are read from
  // `mclamp` (any valid row) and written as zeros
  __device__ __forceinline__ void load_part(const SpgOperand& d, const SpgQuad& q, long mchunk, long mend, long mclamp,
                                            int c0, int i) {
    const int tid = threadIdx.x;
    if (i == 0) vmask = 0;
    const long mm = mchunk + row_of(tid, i);
    const bool ok = mm < mend && (MODE != SPG_PRO_CLOUD || (i == 0 && quad_of(tid) < QUADS));
    vmask |= ok ? (1u << i) : 0u;
    const int c = c0 + 4 * quad_of(tid);
    spg_load_raw1<MODE>(d, ok ? mm : mclamp, q.nvalid > 0 ? c : 0, q.nvalid, raw[i]);
  }
  __device__ __forceinline__ void store_part(const SpgQuad& q, float* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    if (MODE == SPG_PRO_CLOUD && (i > 0 || quad_of(tid) >= QUADS)) return;
    *reinterpret_cast<f32x4*>(lds + row_of(tid, i) * (CH + 4) + 4 * quad_of(tid)) =
        spg_finish_raw<MODE>(q, raw[i], (vmask >> i) & 1u);
  }
};

template <int JT>
struct SpgWeightPipe {        // out-major [JT x 32] weight tile: W [nout, kred]
  static constexpr int NI = JT / 32;
  f32x4 raw[NI];
  unsigned vmask;
  __device__ __forceinline__ void load(const float* __restrict__ W, long ld, int n0, int nout, int k0, int kred) {
    const int tid = threadIdx.x, k = k0 + 4 * (tid & 7);
    vmask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = (tid >> 3) + 32 * i;
      const bool ok = n0 + j < nout && k < kred;
      raw[i] = *reinterpret_cast<const f32x4*>(W + (ok ? (long)(n0 + j) * ld + k : 0));   // unconditional, clamped
      vmask |= ok ? (1u << i) : 0u;       // applied in store(): nothing consumes the loads before the MFMAs
    }
  }
  __device__ __forceinline__ void store(f32x4* __restrict__ lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x4 v = raw[i];
      const bool ok = (vmask >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
      lds[(tid & 7) * (JT + 1) + (tid >> 3) + 32 * i] = v;
    }
  }
  __device__ __forceinline__ void load_part(const float* __restrict__ W, long ld, int n0, int nout, int k0, int kred, int i) {
    const int tid = threadIdx.x, k = k0 + 4 * (tid & 7), j = (tid >> 3) + 32 * i;
    if (i == 0) vmask = 0;
    const bool ok = n0 + j < nout && k < kred;
    raw[i] = *reinterpret_cast<const f32x4*>(W + (ok ? (long)(n0 + j) * ld + k : 0));
    vmask |= ok ? (1u << i) : 0u;
  }
  __device__ __forceinline__ void store_part(f32x4* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    f32x4 v = raw[i];
    const bool ok = (vmask >> i) & 1u;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
    lds[(tid & 7) * (JT + 1) + (tid >> 3) + 32 * i] = v;
  }
};

template <int JT>
struct SpgWeightRedPipe {     // red-major [32 x JT] weight tile: W [kred, nout] read untransposed
  static constexpr int QUADS = JT / 4, RPP = SPG_THREADS / QUADS, NI = SPG_KC / RPP;
  f32x4 raw[NI];
  unsigned vmask;
  __device__ __forceinline__ void load(const float* __restrict__ W, long ld, int n0, int nout, int k0, int kred) {
    const int tid = threadIdx.x, c = n0 + 4 * (tid % QUADS);
    vmask = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / QUADS + RPP * i;
      const bool ok = k0 + r < kred && c < nout;
      raw[i] = *reinterpret_cast<const f32x4*>(W + (ok ? (long)(k0 + r) * ld + c : 0));
      vmask |= ok ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x4 v = raw[i];
      const bool ok = (vmask >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(lds + (tid / QUADS + RPP * i) * (JT + 4) + 4 * (tid % QUADS)) = v;
    }
  }
  __device__ __forceinline__ void load_part(const float* __restrict__ W, long ld, int n0, int nout, int k0, int kred, int i) {
    const int tid = threadIdx.x, c = n0 + 4 * (tid % QUADS), r = tid / QUADS + RPP * i;
    if (i == 0) vmask = 0;
    const bool ok = k0 + r < kred && c < nout;
    raw[i] = *reinterpret_cast<const f32x4*>(W + (ok ? (long)(k0 + r) * ld + c : 0));
    vmask |= ok ? (1u << i) : 0u;
  }
  __device__ __forceinline__ void store_part(float* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    f32x4 v = raw[i];
    const bool ok = (vmask >> i) & 1u;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
    *reinterpret_cast<f32x4*>(lds + (tid / QUADS + RPP * i) * (JT + 4) + 4 * (tid % QUADS)) = v;
  }
};

// ----------------------------------------------------------------------------------------------
// Split-bf16 arithmetic (opt-in precision modes of the wide row-GEMMs, spg_tune key 7 / SpgGemmParams::prec):
//   prec 3  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   with x_hi = bf16(x), x_lo = bf16(x - x_hi): three
//           v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x
//           the matrix rate, products exact to ~2^-16 relative (the dropped a_lo*b_lo term and the 16-bit residuals),
//           fp32 accumulation as before;
//   prec 1  a_hi*b_hi only: plain bf16 operands (2^-9 relative), fp32 accumulation.
// LDS layout: the SAME planes of 16-byte slots as the fp32 out-major tile -- plane p, slot row -- but a slot holds 8
// consecutive reduction indices as bf16: planes 0-3 the hi parts of k = 8p .. 8p+7, planes 4-7 the lo parts.  A lane
// (row r, half h) reads plane 2s+h (hi) / 4+2s+h (lo) with one ds_read_b128 per MFMA operand (s = k-step of 16).
// ----------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 4 floats -> 4 bf16 hi (2 dwords) + 4 bf16 lo (2 dwords); 12 VALU instructions
__device__ __forceinline__ void spg_split_bf16(const f32x4& v, u32x2& hi, u32x2& lo) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const f32x2 x = {v[2 * e], v[2 * e + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 r = {x[0] - __builtin_bit_cast(float, hp << 16), x[1] - __builtin_bit_cast(float, hp & 0xffff0000u)};
    hi[e] = hp;
    lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  }
}

// out-major tile of an A operand in the bf16 layout: called by the fast rows pipe with its finished float4
// (k = 4*kq .. 4*kq+3 of `row`): hi -> plane kq/2, lo -> plane 4 + kq/2, byte (kq & 1) * 8 of the slot
template <int PREC>
__device__ __forceinline__ void spg_store_bf16_quad(f32x4* __restrict__ lds, int stride, int kq, int row, const f32x4& v) {
  u32x2 hi, lo;
  spg_split_bf16(v, hi, lo);
  u32x2* base = reinterpret_cast<u32x2*>(lds);
  base[2 * ((kq >> 1) * stride + row) + (kq & 1)] = hi;
  if (PREC == 3) base[2 * ((4 + (kq >> 1)) * stride + row) + (kq & 1)] = lo;
}

// weights in the bf16 layout, pre-split in global memory by spg_split_weights_kernel: Wb [2][nout][ldb] bf16 (hi matrix,
// then lo matrix; reduction index contiguous).  One 16-byte unit (8 reduction indices of one output channel) per load.
template <int JT, int PREC>
struct SpgWeightBf16 {
  static constexpr int UNITS = JT * 4 * (PREC == 3 ? 2 : 1), NI = (UNITS + SPG_THREADS - 1) / SPG_THREADS;
  f32x4 raw[NI];
  unsigned voff[NI];
  // unit u: k8 = u & 3, j = (u >> 2) % JT, part = u / (4 * JT) (0 hi, 1 lo)
  __device__ __forceinline__ void init(long ldb, int n0, int nout, long part_stride_bytes) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned u = threadIdx.x + SPG_THREADS * i, uu = u < (unsigned)UNITS ? u : 0u;
      const unsigned k8 = uu & 3u, j = (uu >> 2) % JT, part = uu / (4u * JT);
      voff[i] = (unsigned)(part * part_stride_bytes) + (((int)(n0 + j) < nout ? j : 0u) * (unsigned)ldb + 8u * k8) * 2u;
    }
  }
  __device__ __forceinline__ void load_part(const void* __restrict__ Wb, long ldb, int n0, int k0, int i) {
    raw[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wb) + ((long)n0 * ldb + k0) * 2 + voff[i]);
  }
  __device__ __forceinline__ void store_part(f32x4* __restrict__ lds, int i) const {
    const unsigned u = threadIdx.x + SPG_THREADS * i;
    if (UNITS % SPG_THREADS != 0 && u >= (unsigned)UNITS) return;
    const unsigned k8 = u & 3u, j = (u >> 2) % JT, part = u / (4u * JT);
    lds[(4 * part + k8) * (JT + 1) + j] = raw[i];
  }
};

__device__ __forceinline__ void spg_split_bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const f32x2 x = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// slot of a channel inside a plane of the weight-gradient tiles: the 16 slots of a 256-byte LDS row are rotated by the
// row's index, so that the writes of one instruction (32 channel quads -> channels 4q + e) spread over all banks
// (unrotated: 8-way conflicts), while 16 consecutive channels -- one pass of a ds_read_b128 -- stay on 16 distinct slots
__device__ __forceinline__ int spg_swz(int c) { return (c & ~15) | (((c & 15) + (c >> 4)) & 15); }

// MFMAs of one reduction chunk (K = 32 = two k-steps of 16) from bf16 tiles; `piece(slot)`, slot 0..15, as in
// spg_mfma_chunk_il (the staging work of the next chunks, spread between the MFMAs)
template <int TI, int TJ, int PREC, class Piece>
__device__ __forceinline__ void spg_mfma_chunk_bf16_il(const f32x4* __restrict__ As, const f32x4* __restrict__ Bs, int strideA,
                                                       int strideB, const int (&sa)[TI], const int (&sb)[TJ], int h,
                                                       f32x16 (&acc)[TI][TJ], Piece&& piece) {
  constexpr int NG = 2 * TI * TJ * (PREC == 3 ? 3 : 1);      // MFMA count; the 16 piece slots are spread over them
  int slot = 0, g = 0;
  auto after = [&]() __attribute__((always_inline)) {
    ++g;
    const int upto = (16 * g) / NG;
#pragma unroll
    for (; slot < upto; ++slot) piece(slot);
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // order hi*hi, hi*lo, then lo*hi: the lo fragments of A are loaded into registers only after the B lo fragments are
    // dead (24 fragment registers live instead of 32 -- the backward instantiations have none to spare)
    bf16x8 ah[TI], bh[TJ], xl[TI > TJ ? TI : TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) ah[i] = __builtin_bit_cast(bf16x8, As[(2 * s + h) * strideA + sa[i]]);
#pragma unroll
    for (int j = 0; j < TJ; ++j) bh[j] = __builtin_bit_cast(bf16x8, Bs[(2 * s + h) * strideB + sb[j]]);
    if (PREC == 3) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) xl[j] = __builtin_bit_cast(bf16x8, Bs[(4 + 2 * s + h) * strideB + sb[j]]);
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        after();
      }
    if (PREC == 3) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], xl[j], acc[i][j], 0, 0, 0);
          after();
        }
#pragma unroll
      for (int i = 0; i < TI; ++i) xl[i] = __builtin_bit_cast(bf16x8, As[(4 + 2 * s + h) * strideA + sa[i]]);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], bh[j], acc[i][j], 0, 0, 0);
          after();
        }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Full-tile fast path.  VALU instructions are NOT free next to MFMAs on gfx950 -- they take issue cycles from the same
// SIMD (measured, tools/probe/mfma_probe.hip: 4 VALU per fp32 MFMA cost 25 % of the MFMA rate) -- and in the masked
// pipes above most of the VALU work of a chunk is 64-bit address arithmetic, clamps and validity selects.  When every
// tile of the launch is full (host-side check: no partial rows / channels / reduction chunks), all of that is loop
// invariant: each thread keeps 32-bit byte offsets, the per-chunk base is wave-uniform (scalar unit), loads are
// `global_load_dwordx4 v, v_off, s[base]`, and the prologue arithmetic is the only VALU work left in the loop.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 spg_ld16(const float* __restrict__ ubase, unsigned voff) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ubase) + voff);
}
__device__ __forceinline__ int4 spg_ld16i(const int* __restrict__ ubase, unsigned voff) {
  return *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(ubase) + voff);
}

// shared finish arithmetic of the fast pipes (no masks)
template <int MODE>
__device__ __forceinline__ f32x4 spg_finish_fast(const SpgQuad& q, const SpgRaw& r, float lo, const f32x4& px,
                                                 const int4& pai, int pp) {
  f32x4 v;
  if (MODE == SPG_PRO_IDENT) {
    v = r.x;
  } else if (MODE == SPG_PRO_AFFINE) {
    const f32x4 t = spg_fma4(r.x, q.a, q.b);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(t[e], lo);
  } else if (MODE == SPG_PRO_BNBWD) {
    v = spg_bnbwd_value4(q.a, r.x, q.b, r.y, q.c, q.d);
  } else {   // POOLBWD: the pooled gradient goes to the arg-max point of each (group, channel)
    const f32x4 g = {pai.x == pp ? px[0] : 0.f, pai.y == pp ? px[1] : 0.f, pai.z == pp ? px[2] : 0.f, pai.w == pp ? px[3] : 0.f};
    v = spg_bnbwd_value4(q.a, g, q.b, r.y, q.c, q.d);
  }
  return v;
}

template <int MODE>
__device__ __forceinline__ void spg_consts_fast(const SpgOperand& d, int c, unsigned coff, SpgQuad& q) {
  if (MODE == SPG_PRO_AFFINE) {
    q.a = spg_ld16(d.c0 + c, coff);
    q.b = spg_ld16(d.c1 + c, coff);
  } else if (MODE == SPG_PRO_BNBWD || MODE == SPG_PRO_POOLBWD) {
    q.a = spg_ld16(d.c0 + c, coff); q.b = spg_ld16(d.c1 + c, coff);
    q.c = spg_ld16(d.c2 + c, coff); q.d = spg_ld16(d.c3 + c, coff);
  }
}

template <int MODE, int ROWS>
struct SpgRowsFast {           // out-major [ROWS x 32] tile, every row and channel valid; POOLBWD: the tile is one group
  static constexpr int NI = ROWS / 32;
  SpgRaw raw[NI];
  SpgQuad q;
  f32x4 px;                    // POOLBWD: pooled gradient / arg-max of the group, channels of this thread's quad
  int4 pai;
  unsigned voff[NI], coff;
  float lo;
  // rows >= mvalid (partial last tile) read row 0 of the tile: harmless garbage, their outputs are never stored
  __device__ __forceinline__ void init(const SpgOperand& d, int mvalid) {
    const unsigned tid = threadIdx.x, kq = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned row = r0 + 32u * i;
      voff[i] = ((row < (unsigned)mvalid ? row : 0u) * (unsigned)d.ld + 4u * kq) * 4u;
    }
    coff = 16u * kq;
    lo = d.relu ? 0.f : -FLT_MAX;
  }
  __device__ __forceinline__ void prepare(const SpgOperand& d, long group, int k0) {
    spg_consts_fast<MODE>(d, k0, coff, q);
    if (MODE == SPG_PRO_POOLBWD) {
      px = spg_ld16(d.X + group * d.ldg + k0, coff);
      pai = spg_ld16i(d.aidx + group * d.ldg + k0, coff);
    }
  }
  __device__ __forceinline__ void load_part(const SpgOperand& d, long m0, int k0, int i) {
    if (MODE == SPG_PRO_POOLBWD) {
      raw[i].y = spg_ld16(d.X2 + m0 * d.ld + k0, voff[i]);
    } else {
      raw[i].x = spg_ld16(d.X + m0 * d.ld + k0, voff[i]);
      if (MODE == SPG_PRO_BNBWD) raw[i].y = spg_ld16(d.X2 + m0 * d.ld + k0, voff[i]);
    }
  }
  __device__ __forceinline__ void store_part(f32x4* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    lds[(tid & 7) * (ROWS + 1) + (tid >> 3) + 32 * i] = spg_finish_fast<MODE>(q, raw[i], lo, px, pai, (tid >> 3) + 32 * i);
  }
  template <int PREC>
  __device__ __forceinline__ void store_part_bf16(f32x4* __restrict__ lds, int i) const {      // bf16 layout (see spg_store_bf16_quad)
    const int tid = threadIdx.x;
    spg_store_bf16_quad<PREC>(lds, ROWS + 1, tid & 7, (tid >> 3) + 32 * i, spg_finish_fast<MODE>(q, raw[i], lo, px, pai, (tid >> 3) + 32 * i));
  }
};

template <int JT>
struct SpgWeightFast {         // out-major [JT x 32] weight tile, W [nout, kred], all valid
  static constexpr int NI = JT / 32;
  f32x4 raw[NI];
  unsigned voff[NI];
  // output channels >= nout (partial last column tile) read channel n0: their columns are never stored
  __device__ __forceinline__ void init(long ld, int n0, int nout) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned j = (tid >> 3) + 32u * i;
      voff[i] = (((int)(n0 + j) < nout ? j : 0u) * (unsigned)ld + 4u * (tid & 7)) * 4u;
    }
  }
  __device__ __forceinline__ void load_part(const float* __restrict__ W, long ld, int n0, int k0, int i) {
    raw[i] = spg_ld16(W + (long)n0 * ld + k0, voff[i]);
  }
  __device__ __forceinline__ void store_part(f32x4* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    lds[(tid & 7) * (JT + 1) + (tid >> 3) + 32 * i] = raw[i];
  }
};

template <int JT>
struct SpgWeightRedFast {      // red-major [32 x JT] weight tile, W [kred, nout] read untransposed, all valid
  static constexpr int QUADS = JT / 4, RPP = SPG_THREADS / QUADS, NI = SPG_KC / RPP;
  f32x4 raw[NI];
  unsigned voff[NI];
  __device__ __forceinline__ void init(long ld, int n0, int nout) {
    const unsigned tid = threadIdx.x;
    const unsigned cq = 4u * (tid % QUADS);
#pragma unroll
    for (int i = 0; i < NI; ++i) voff[i] = ((tid / QUADS + RPP * i) * (unsigned)ld + ((int)(n0 + cq) < nout ? cq : 0u)) * 4u;
  }
  __device__ __forceinline__ void load_part(const float* __restrict__ W, long ld, int n0, int k0, int i) {
    raw[i] = spg_ld16(W + (long)k0 * ld + n0, voff[i]);
  }
  __device__ __forceinline__ void store_part(float* __restrict__ lds, int i) const {
    const int tid = threadIdx.x;
    *reinterpret_cast<f32x4*>(lds + (tid / QUADS + RPP * i) * (JT + 4) + 4 * (tid % QUADS)) = raw[i];
  }
};

template <int MODE, int CH>
struct SpgRedFast {            // red-major [32 x CH] operand tile of the weight gradient, all rows / channels valid
  static constexpr int QUADS = CH / 4, RPP = SPG_THREADS / QUADS, NI = SPG_KC / RPP;
  SpgRaw raw[NI];
  f32x4 px;
  int4 pai;
  unsigned voff[NI], coff;
  int pbase;                   // POOLBWD: point index of this thread's first row relative to the chunk start
  float lo;
  __device__ __forceinline__ void init(const SpgOperand& d, int c0, SpgQuad& q) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) voff[i] = ((tid / QUADS + RPP * i) * (unsigned)d.ld + 4u * (tid % QUADS)) * 4u;
    coff = 16u * (tid % QUADS);
    lo = d.relu ? 0.f : -FLT_MAX;
    spg_consts_fast<MODE>(d, c0, coff, q);
  }
  // the 32 rows of a chunk belong to ONE pooling group (P % 32 == 0, chunk starts are multiples of 32)
  __device__ __forceinline__ void load_part(const SpgOperand& d, long mchunk, int c0, int i) {
    if (MODE == SPG_PRO_POOLBWD) {
      if (i == 0) {
        const unsigned g = (unsigned)mchunk / (unsigned)d.P;      // M < 2^32 rows
        pbase = (int)((unsigned)mchunk - g * (unsigned)d.P);
        px = spg_ld16(d.X + g * d.ldg + c0, coff);
        pai = spg_ld16i(d.aidx + g * d.ldg + c0, coff);
      }
      raw[i].y = spg_ld16(d.X2 + mchunk * d.ld + c0, voff[i]);
    } else {
      raw[i].x = spg_ld16(d.X + mchunk * d.ld + c0, voff[i]);
      if (MODE == SPG_PRO_BNBWD) raw[i].y = spg_ld16(d.X2 + mchunk * d.ld + c0, voff[i]);
    }
  }
  __device__ __forceinline__ void store_part(const SpgQuad& q, float* __restrict__ lds, int i) const {
    const int tid = threadIdx.x, row = tid / QUADS + RPP * i;
    *reinterpret_cast<f32x4*>(lds + row * (CH + 4) + 4 * (tid % QUADS)) =
        spg_finish_fast<MODE>(q, raw[i], lo, px, pai, pbase + row);
  }
};

// bf16-layout variant of SpgRedFast for the weight gradient (reduction = rows): the MFMA operand of channel c needs 8
// CONSECUTIVE ROWS of c in one 16-byte slot, the global rows hold consecutive channels -- so a thread takes NI consecutive
// rows of its channel quad (instead of every RG-th row), finishes them, and for each of its 4 channels packs the NI values
// into NI*2 bytes of the slot (plane = first row / 8): an in-register transpose, no shuffles.
template <int MODE, int CH>
struct SpgRedFastB {
  static constexpr int QUADS = CH / 4, RG = SPG_THREADS / QUADS, NI = SPG_KC / RG;      // NI = 4 / 2 / 1 for CH = 128 / 64 / 32
  SpgRaw raw[NI];
  f32x4 px;
  int4 pai;
  unsigned voff[NI], coff, wofs[4];
  int pbase, row0;
  float lo;
  __device__ __forceinline__ void init(const SpgOperand& d, int c0, SpgQuad& q) {
    const unsigned tid = threadIdx.x, quad = tid % QUADS;
    row0 = (int)(tid / QUADS) * NI;
#pragma unroll
    for (int i = 0; i < NI; ++i) voff[i] = ((unsigned)(row0 + i) * (unsigned)d.ld + 4u * quad) * 4u;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      wofs[e] = (unsigned)(((row0 >> 3) * (CH + 1) + spg_swz((int)(4 * quad + e))) * 16 + (row0 & 7) * 2);
    coff = 16u * quad;
    lo = d.relu ? 0.f : -FLT_MAX;
    spg_consts_fast<MODE>(d, c0, coff, q);
  }
  __device__ __forceinline__ void load_part(const SpgOperand& d, long mchunk, int c0, int i) {
    if (MODE == SPG_PRO_POOLBWD) {
      if (i == 0) {
        const unsigned g = (unsigned)mchunk / (unsigned)d.P;      // M < 2^32 rows; a chunk lies inside one pooling group
        pbase = (int)((unsigned)mchunk - g * (unsigned)d.P);
        px = spg_ld16(d.X + g * d.ldg + c0, coff);
        pai = spg_ld16i(d.aidx + g * d.ldg + c0, coff);
      }
      raw[i].y = spg_ld16(d.X2 + mchunk * d.ld + c0, voff[i]);
    } else {
      raw[i].x = spg_ld16(d.X + mchunk * d.ld + c0, voff[i]);
      if (MODE == SPG_PRO_BNBWD) raw[i].y = spg_ld16(d.X2 + mchunk * d.ld + c0, voff[i]);
    }
  }
  // channels 2*half, 2*half+1 of the thread's quad (two calls per chunk: the conversion work is spread over two slots)
  template <int PREC>
  __device__ __forceinline__ void store_half(const SpgQuad& q, f32x4* __restrict__ lds, int half) const {
    f32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = spg_finish_fast<MODE>(q, raw[i], lo, px, pai, pbase + row0 + i);
    char* base = reinterpret_cast<char*>(lds);
    constexpr unsigned LO = 4u * (CH + 1) * 16u;
#pragma unroll
    for (int ee = 0; ee < 2; ++ee) {
      const int e = 2 * half + ee;
      if constexpr (NI == 4) {
        unsigned h0, l0, h1, l1;
        spg_split_bf16_pair(v[0][e], v[1][e], h0, l0);
        spg_split_bf16_pair(v[2][e], v[3][e], h1, l1);
        *reinterpret_cast<u32x2*>(base + wofs[e]) = u32x2{h0, h1};
        if (PREC == 3) *reinterpret_cast<u32x2*>(base + wofs[e] + LO) = u32x2{l0, l1};
      } else if constexpr (NI == 2) {
        unsigned hi, lw;
        spg_split_bf16_pair(v[0][e], v[1][e], hi, lw);
        *reinterpret_cast<unsigned*>(base + wofs[e]) = hi;
        if (PREC == 3) *reinterpret_cast<unsigned*>(base + wofs[e] + LO) = lw;
      } else {
        unsigned hi, lw;
        spg_split_bf16_pair(v[0][e], 0.f, hi, lw);
        *reinterpret_cast<unsigned short*>(base + wofs[e]) = (unsigned short)hi;
        if (PREC == 3) *reinterpret_cast<unsigned short*>(base + wofs[e] + LO) = (unsigned short)lw;
      }
    }
  }
};

// C/D fragment of v_mfma_f32_32x32x2_f32: lane holds column (lane&31), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int spg_acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Sum over the 64 lanes of a wavefront, the total in every lane.  Data-parallel-primitive moves on the vector ALU (a few
// cycles each) instead of six dependent LDS-crossbar shuffles (ds_bpermute, ~100 cycles each): the per-node kernels run one
// wave per SIMD, so the length of the dependent chain is the run time.  Steps: pairs (quad_perm 1,0,3,2), quads (2,3,0,1),
// eights (row_half_mirror -- the quads are uniform by then, so mirroring equals xor 4), sixteens (row_mirror), then the row
// totals travel up the rows (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3); lane 63 holds the total, which
// is broadcast through a scalar register.  Fixed association order: deterministic.
__device__ __forceinline__ float spg_dpp_add(float v, float src, int ctrl, int row_mask) {
  // v + (src moved by the DPP pattern; lanes of rows outside row_mask receive 0)
  int t;
  switch (ctrl) {      // the control word must be an immediate
    case 0xB1: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0xB1, 0xF, 0xF, false); break;
    case 0x4E: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x4E, 0xF, 0xF, false); break;
    case 0x141: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x141, 0xF, 0xF, false); break;
    case 0x140: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x140, 0xF, 0xF, false); break;
    case 0x142: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x142, 0xA, 0xF, false); break;
    default: t = __builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x143, 0xC, 0xF, false); break;
  }
  (void)row_mask;
  return v + __int_as_float(t);
}
__device__ __forceinline__ float spg_wave_sum(float v) {
  v = spg_dpp_add(v, v, 0xB1, 0xF);
  v = spg_dpp_add(v, v, 0x4E, 0xF);
  v = spg_dpp_add(v, v, 0x141, 0xF);
  v = spg_dpp_add(v, v, 0x140, 0xF);
  v = spg_dpp_add(v, v, 0x142, 0xA);
  v = spg_dpp_add(v, v, 0x143, 0xC);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
