// Fused row-GEMM kernels for gfx950: the dense contractions of the PointNet 1x1 convolutions / FC layers
// and of the filter-generating network (reference: learning/pointnet.py:27-49,83-118,
// learning/graphnet.py:17-34), forward, data-gradient and weight-gradient, on v_mfma_f32_32x32x2_f32
// (exact fp32, equal to an fmaf chain) with
//   * the producer's BatchNorm+ReLU (forward) / BatchNorm-backward formula (backward) fused into the
//     LDS staging of the consumer, so normalised activations never exist in HBM;
//   * per-workgroup BatchNorm partial statistics (mean / M2 for Chan's combination; sum dz, sum dz*xhat
//     in the backward), the max-pool over the points of a superpoint and the ReLU mask fused into the
//     epilogue.
// One workgroup = 4 wavefronts (one per SIMD); tile = 128 rows x JT columns, reduction staged through
// LDS in chunks of 32 in the [k/4][row][4] layout of spg_common.h.
// No IMPLICIT fused multiply-add contraction in this translation unit (explicit fmaf only): whether a * b + c written as two
// operations becomes one fma is the compiler's choice and depends on the surrounding code, so the same kernel body inlined
// into two kernels -- its stand-alone launch and the grouped launch (spg_multi_kernel) -- could round differently.  The bodies
// spell out the fmas they want; everything else rounds the same in every context (tests/test_gpu_grouped.py: bit-identical).
#pragma clang fp contract(off)
#include "../../include/spg_hip.h"
#include "spg_gemm.h"
#include <type_traits>
#include <float.h>
#include <limits.h>
#include <stdarg.h>

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" const char* spg_last_error(void) { return g_err; }
void spg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// optional instrumentation (bench.py): hipEvents around every MFMA GEMM launch, on the launch stream
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
#ifdef SPG_ATTRIBUTION
#define SPG_DBG(p) ((p).dbg)
#else
#define SPG_DBG(p) 0
#endif
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int tag; int M, N, K; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mutex;      // host threads may launch on different streams while the instrumentation is on
struct ProfScope {
  hipStream_t st; bool on; ProfRec r;
  ProfScope(hipStream_t s, double flops, int tag = 0, bool enable = true) : st(s), on(g_prof_on && enable) {
    if (on) { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); r.flops = flops; r.tag = tag; r.M = r.N = r.K = 0; (void)hipEventRecord(r.a, st); }
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(r.b, st); std::lock_guard<std::mutex> lock(g_prof_mutex); g_prof.push_back(r); } }
};
}  // namespace
extern "C" void spg_prof_enable(int on) { g_prof_on = on != 0; }
// the instrumentation for launchers outside this file (spg_gemm.h: SpgProfSpan)
SpgProfSpan::SpgProfSpan(hipStream_t stream, double flops, int tag, int N, int K) : impl_(nullptr) {
  if (!g_prof_on) return;
  ProfScope* ps = new ProfScope(stream, flops, tag);
  ps->r.N = N; ps->r.K = K;
  impl_ = ps;
}
SpgProfSpan::~SpgProfSpan() { delete static_cast<ProfScope*>(impl_); }

// tuning knobs (spg_tune): process-global, read by the launchers
// Timing-attribution switches (spg_tune key 3: parts of the persistent forward kernel switched off, results WRONG) and the
// per-tile statistics switch (key 5) exist only in a build with -DSPG_ATTRIBUTION (`make ATTRIBUTION=1`, used by
// tools/tune_sweep.py and the r02 attribution records); the production library compiles them to constants.
static int g_tune[SPG_TUNE_COUNT] = {0};
static int spg_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}
int spg_tune_get(int key) { return (key >= 0 && key < SPG_TUNE_COUNT) ? g_tune[key] : 0; }
extern "C" int spg_tune(int key, int value) {
  if (key < 0 || key >= SPG_TUNE_COUNT) return -1;
#ifndef SPG_ATTRIBUTION
  if ((key == SPG_TUNE_DBG || key == SPG_TUNE_NO_STAT_ACCUM) && value != 0) return -1;      // not in the production build
#endif
  const int old = g_tune[key];
  g_tune[key] = value;
  return old;
}

namespace { thread_local int g_riding = 0; }
struct SpgRidingScope { SpgRidingScope() { ++g_riding; } ~SpgRidingScope() { --g_riding; } };
#define SPG_OWNER_WEIGHT 100000

// per-(instantiation, shape) totals of the instrumented launches: up to `max` rows {tag, N, K, launches} / {ms, flops}
extern "C" int spg_prof_read_shapes(int* keys, double* vals, int max) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  int n = 0;
  for (ProfRec& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float dt = 0.f;
    (void)hipEventElapsedTime(&dt, r.a, r.b);
    int j = 0;
    for (; j < n; ++j)
      if (keys[4 * j] == r.tag && keys[4 * j + 1] == r.N && keys[4 * j + 2] == r.K) break;
    if (j == n) {
      if (n == max) continue;
      keys[4 * n] = r.tag; keys[4 * n + 1] = r.N; keys[4 * n + 2] = r.K; keys[4 * n + 3] = 0;
      vals[2 * n] = 0.0; vals[2 * n + 1] = 0.0;
      ++n;
    }
    keys[4 * j + 3] += 1; vals[2 * j] += dt; vals[2 * j + 1] += r.flops;
  }
  return n;
}
// kernel tag of an instrumented launch: kind (1 row-GEMM, 2 weight gradient), tile, weight layout / operand modes, fast path
#define SPG_PROF_TAG(kind, IT, JT, X, Y, FULL) ((kind) * 1000000 + ((IT) / 32) * 100000 + ((JT) / 32) * 10000 + ((X) + 1) * 100 + ((Y) + 1) * 10 + (FULL))
extern "C" int spg_prof_tag(int kind, int it, int jt, int x, int y, int full) { return SPG_PROF_TAG(kind, it, jt, x, y, full ? 1 : 0); }
extern "C" int spg_prof_read_tag(int tag, double* ms, long* launches, double* flops) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  double t = 0.0, f = 0.0;
  long n = 0;
  for (ProfRec& r : g_prof) {
    if (r.tag != tag) continue;
    (void)hipEventSynchronize(r.b);
    float dt = 0.f;
    (void)hipEventElapsedTime(&dt, r.a, r.b);
    t += dt; f += r.flops; ++n;
  }
  if (ms) *ms = t;
  if (launches) *launches = n;
  if (flops) *flops = f;
  return 0;
}
extern "C" int spg_prof_read(double* ms, long* launches, double* flops, int reset) {
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  double t = 0.0, f = 0.0;
  for (ProfRec& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float dt = 0.f;
    (void)hipEventElapsedTime(&dt, r.a, r.b);
    t += dt; f += r.flops;
  }
  if (ms) *ms = t;
  if (launches) *launches = (long)g_prof.size();
  if (flops) *flops = f;
  if (reset) {
    for (ProfRec& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// grouped launches (spg_gemm.h: SpgGroupScope): job table shared by the launchers and spg_multi_kernel (end of this file)
// ---------------------------------------------------------------------------------------------
#include "spg_ecc.h"
enum { SPG_JOB_GEMM = 1, SPG_JOB_WGRAD = 2, SPG_JOB_COLSUM = 3, SPG_JOB_EDGE_WGRAD = 4, SPG_JOB_PAD_ROWS = 5, SPG_JOB_ZERO = 6, SPG_JOB_REDUCE = 7 };
struct SpgSmallJob {            // column sums: src [M, ld] -> dst [slices][N] (a = ld, b = M, c = rows per slice, i = N)
  const float* src;             // pad rows:    src [rows, a] -> dst [rows, b] zero padded (c = rows, i = cols)
  float* dst;                   // zero:        dst [b floats]
  long a, b, c;
  int i;
};
// Kernel arguments of a grouped launch (<= 4 KiB): a table of job headers in launch order + the jobs' parameter structs packed
// back to back (16-byte aligned) in one byte arena -- 12 small jobs or e.g. 2 row-GEMMs + 4 weight gradients + 4 small ones.
struct SpgJobHdr { int kind, variant, gx, gy, gz, offset, weight, bx0; };      // bx0: first x index of the job's grid (a SLICE of a larger launch)
#define SPG_GROUP_MAX_JOBS 16
#define SPG_GROUP_ARENA_BYTES 3440
#define SPG_GROUP_MAX_WGRAD_ROWS 65536      // weight gradients over more rows than this are never grouped
struct SpgMultiArgs {
  int njobs, pad;
  unsigned long long* trace;                // attribution build: per-job [min start, max end] of this launch (wall_clock64), or null
  int first_block[SPG_GROUP_MAX_JOBS];      // first workgroup of every job: ONE batch of scalar loads finds a workgroup's job
  SpgJobHdr hdr[SPG_GROUP_MAX_JOBS];
  alignas(16) unsigned char arena[SPG_GROUP_ARENA_BYTES];
};
static_assert(sizeof(SpgMultiArgs) <= 4096, "kernel arguments of a grouped launch must fit 4 KiB");
static_assert(sizeof(SpgGemmParams) % 8 == 0 && sizeof(SpgWgradParams) % 8 == 0, "parameter structs are packed back to back");
// few-row row-GEMM bodies spg_rowgemm_body<32, 128, 1, 4, WRED, AMODE, FULL> the group can run: id, or -1
constexpr int spg_gemm_variant(bool wred, int amode, bool full) {
  const int m = amode == -1 ? (full ? -1 : 4)
              : amode == SPG_PRO_IDENT ? (full ? 1 : 0)
              : ((!wred && amode == SPG_PRO_AFFINE) || (wred && amode == SPG_PRO_BNBWD)) ? (full ? 3 : 2) : -1;
  return m < 0 ? -1 : (wred ? 5 : 0) + m;
}
// Two builds of the grouped kernel: LIGHT (bodies that fit 128 registers: 4 workgroups per CU -- what a group of MANY small
// workgroups needs, e.g. the per-edge filter gradient next to the cell's weight gradients) and HEAVY (every body, 2 per CU).
// A group is launched with the light build unless one of its jobs is a heavy variant (register counts of the stand-alone
// instantiations, -Rpass-analysis=kernel-resource-usage).
constexpr bool spg_gemm_variant_light(int v) { return v != 8; }      // 8 = data gradient, BNBWD prologue, full tiles: 144 VGPRs
// split-K few-row bodies spg_fewrow_sk_body<WRED, AMODE>: ids 10 .. 13
constexpr int spg_gemm_sk_variant(bool wred, int amode) {
  return amode == SPG_PRO_IDENT ? (wred ? 12 : 10) : ((!wred && amode == SPG_PRO_AFFINE) ? 11 : ((wred && amode == SPG_PRO_BNBWD) ? 13 : -1));
}
// weight-gradient bodies the group can run (the shapes the few-row layers of the S3DIS / Semantic3D configurations produce)
//   X(id, IT, JT, WI, WJ, AMODE, BMODE, FULL, COLSUM, LIGHT)
#define SPG_WGRAD_VARIANTS(X)                                                                                    \
  X(0, 128, 128, 2, 2, 3, 1, false, false, false)   X(1, 128, 128, 2, 2, 3, 1, true, false, false)               \
  X(2, 64, 64, 2, 2, 0, 1, false, true, true)       X(3, 64, 64, 2, 2, 0, 1, true, true, true)                   \
  X(4, 128, 64, 2, 2, 0, 1, false, false, false)    X(5, 128, 64, 2, 2, 0, 1, true, false, false)                \
  X(6, 128, 128, 2, 2, -1, -1, false, true, false)                                                               \
  X(7, 128, 32, 4, 1, 0, 0, false, false, true)     X(8, 128, 32, 4, 1, 0, 0, true, false, true)                 \
  X(9, 128, 32, 4, 1, -1, -1, false, true, false)                                                                \
  X(10, 128, 32, 4, 1, 0, 0, false, true, false)    X(11, 128, 32, 4, 1, 0, 0, true, true, true)                 \
  X(12, 128, 32, 4, 1, 0, 1, false, true, false)    X(13, 128, 32, 4, 1, 0, 1, true, true, false)                \
  X(14, 64, 64, 2, 2, 0, 1, false, false, true)     X(15, 64, 64, 2, 2, 0, 1, true, false, true)                 \
  X(16, 128, 128, 2, 2, 0, 1, false, true, false)   X(17, 128, 128, 2, 2, 0, 1, true, true, false)               \
  X(18, 128, 64, 2, 2, 3, 1, false, false, false)   X(19, 128, 64, 2, 2, 3, 1, true, false, false)               \
  /* leaves of PointNet's backward that travel in slices next to the STN head's launches (spg_gemm.h: spg_leaf_*): the      \
     pooled convolution's weight gradient (max-pool scatter x affine) and the first convolution's (BatchNorm-backward x cloud) */ \
  X(20, 128, 128, 2, 2, 4, 1, true, false, false)   X(21, 128, 32, 4, 1, 3, 2, false, false, false)
constexpr int spg_wgrad_variant(int it, int jt, int amode, int bmode, bool full, bool colsum) {
#define SPG_X(id, IT_, JT_, WI_, WJ_, AM_, BM_, FU_, CS_, LI_) \
  if (it == IT_ && jt == JT_ && amode == AM_ && bmode == BM_ && full == FU_ && colsum == CS_) return id;
  SPG_WGRAD_VARIANTS(SPG_X)
#undef SPG_X
  return -1;
}
constexpr bool spg_wgrad_variant_light(int v) {
#define SPG_X(id, IT_, JT_, WI_, WJ_, AM_, BM_, FU_, CS_, LI_) if (v == id) return LI_;
  SPG_WGRAD_VARIANTS(SPG_X)
#undef SPG_X
  return false;
}
// host side (end of this file): true = the job was taken by the group that is open on this thread
static bool spg_group_accepts(hipStream_t stream);
static bool spg_group_add(int kind, int variant, const void* params, size_t bytes, dim3 grid, size_t lds, double flops, hipStream_t stream, int weight,
                          std::function<int()> direct = nullptr, int bx0 = 0);

// ---------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------------
// Epilogues.  FULL: the tile is completely inside the matrix (and, for the backward, inside the masked channel
// range), so every load / store is unconditional: straight-line code without per-element exec-mask branches and the
// conservative s_waitcnt the compiler puts at their joins.
// LDS floats of one wave's staging region for the vector store: 16 rows x (cols + 8) (the +8 keeps the two lane halves of
// an accumulator register on disjoint banks).  The sub-tile goes through it in pieces of 16 rows (= 8 accumulator
// registers per 32x32 block), so that the staging of all four waves fits into ONE of the two main-loop LDS buffers: the
// other one already holds the first chunk of the workgroup's next tile (persistent launches).
// a zero the optimiser cannot see through: added to the thread index inside the epilogues so that their (many) per-lane
// address computations are NOT hoisted out of the persistent tile loop (they would stay live across the main loop)
__device__ __forceinline__ int spg_opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
template <typename T> using spg_kernarg_ptr = __attribute__((address_space(4))) const T*;
// Parameters are read where they lie (scalar loads at the point of use, see spg_rowgemm_kernel): at the start of a kernel every
// 64-byte line of the block is a scalar-cache MISS, and the uses come one after the other -- load, wait, use, next load: nine misses
// in a row for the 536 bytes of SpgGemmParams (tools/fwd_phase_timing.py: 5 400-6 900 cycles from the first instruction to the
// first tile load).  One batch of dummy loads, one wait: the lines arrive together, the later loads hit.
template <int BYTES>
__device__ __forceinline__ void spg_touch_params(const spg_kernarg_ptr<unsigned> b) {
  unsigned acc = 0;
#pragma unroll
  for (int o = 0; o < BYTES; o += 64) acc |= b[o / 4];
  asm volatile("" :: "s"(acc));
}
#define SPG_EPI_PIECE_ROWS 16
#define SPG_EPI_WAVE_FLOATS(RW, CW) (SPG_EPI_PIECE_ROWS * ((CW) + 8))

// accumulators of one wave (RW x CW sub-tile in the MFMA C layout) -> LDS (wave-private, no barrier) -> rows of float4
template <int RW, int CW, int TI, int TJ>
__device__ __forceinline__ void spg_store_tile_vec_impl(const f32x16 (&acc)[TI][TJ], float* __restrict__ st,
                                                        float* __restrict__ ybase, unsigned ldy, int lane) {
  constexpr int LD = CW + 8, LPR = CW / 4, RPI = 64 / LPR;      // lanes per row, rows per store instruction
  static_assert(SPG_EPI_PIECE_ROWS % RPI == 0, "a piece is a whole number of store instructions");
  const int r = lane & 31, h = lane >> 5;
  const int lr = lane / LPR, lc = 4 * (lane % LPR);
  const unsigned o0 = (unsigned)lr * ldy + (unsigned)lc;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // registers 8*half .. 8*half+7 of a 32x32 block are its rows 16*half + (u&3) + 8*(u>>2) + 4h
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) st[((u & 3) + 8 * (u >> 2) + 4 * h) * LD + 32 * j + r] = acc[i][j][8 * half + u];
#pragma unroll
      for (int it = 0; it < SPG_EPI_PIECE_ROWS / RPI; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(st + (lr + RPI * it) * LD + lc);
        // NON-TEMPORAL (round 6): a layer's raw output is read again by the next layer once and then by the backward pass half a
        // millisecond and gigabytes later -- it has no business displacing anything in L2 (same box, interleaved: 1.107 -> 1.101 ms / step)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ybase + o0 + (unsigned)(32 * i + 16 * half + RPI * it) * ldy));
      }
    }
}
template <int RW, int CW, int TI, int TJ>
__device__ __forceinline__ void spg_store_tile_vec(const f32x16 (&acc)[TI][TJ], float* __restrict__ st,
                                                   float* __restrict__ ybase, unsigned ldy, int lane) {
  static_assert(TI * 32 == RW && TJ * 32 == CW, "wave sub-tile");
  spg_store_tile_vec_impl<RW, CW, TI, TJ>(acc, st, ybase, ldy, lane);
}

#include "spg_fold.h"

// Statistics of a persistent workgroup: every wave keeps (rows, mean, M2) of ITS columns over the tiles it has seen
// (merged tile by tile with Chan's formula) and writes ONE partial when the stream ends -- 4x fewer partials for the
// finalize kernel, which then needs no slicing / last-arrival ticket.  Backward: plain running sums.
template <int TJ>
struct SpgStatAcc {
  float n;
  float a[TJ], b[TJ];        // forward: mean, M2 (lane = column); backward (vector epilogue): unused
  f32x4 s1, s2;              // backward vector epilogue: sum dz, sum dz*xhat of the lane's channel quad
  float sg[TJ];              // forward, pooled layer: the BatchNorm scale of the lane's columns (its sign is what the pooling needs;
  bool has_sg;               //   loop-invariant: loaded once by a persistent stream, has_sg says so)
};
// ROUND 6: this struct must stay in REGISTERS.  It used to reach the epilogues as `accum ? &sacc : nullptr`; a conditional address
// defeats scalar replacement, the running statistics lived in scratch memory, and every scratch access is a VECTOR memory operation
// behind an in-order counter: `s_waitcnt vmcnt` in front of the first one waited for the tile's sixteen global stores and for the
// next tile's prefetched loads -- 7 000-12 000 cycles per tile in the "statistics" of a 16 384-cycle tile (tools/fwd_phase_timing.py,
// profiles/r06_fwd_phase_timing.txt).  Now: always a reference, the choice is a separate flag.

// BIAS_DONE: the accumulators were initialised with the bias (persistent fast path), nothing to add here
template <int IT, int JT, int WI, int WJ, bool FULL, bool BIAS_DONE = false>
__device__ __forceinline__ void spg_epilogue_fwd(const SpgGemmParams& p, f32x16 (&acc)[IT / WI / 32][JT / WJ / 32],
                                                 float* __restrict__ red, int tile, long m0, int mvalid, int n0,
                                                 SpgStatAcc<JT / WJ / 32>& sacc, const bool accum, unsigned long long* edbg = nullptr) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32;
  const int tid = threadIdx.x + spg_opaque_zero(), lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  const int colw = wj * (JT / WJ), roww = wi * (IT / WI);
#ifdef SPG_ATTRIBUTION
  unsigned long long et__ = __builtin_readcyclecounter();
#define SPG_EP(k) if (edbg != nullptr) { const unsigned long long n__ = __builtin_readcyclecounter(); edbg[k] += n__ - et__; et__ = n__; }
#else
#define SPG_EP(k)
#endif
  // ---- bias + store ----
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + colw + 32 * j + r;
    const bool colok = FULL || col < p.N;
    if (!BIAS_DONE && p.bias != nullptr) {   // uniform
      const float t = p.bias[colok ? col : 0];
      const float bv = colok ? t : 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] += bv;
    }
    if (p.Y != nullptr && !(FULL && p.vec_store)) {      // uniform
      float* yb = p.Y + m0 * p.ldy + n0;                       // wave-uniform base (SGPRs) + 32-bit lane offsets
      const unsigned ldy = (unsigned)p.ldy;
      const unsigned o0 = (unsigned)(roww + 4 * h) * ldy + (unsigned)(colw + 32 * j + r);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 32 * i + spg_acc_row(q, h);
          if (FULL || (colok && roww + row < mvalid)) yb[o0 + (unsigned)(32 * i + (q & 3) + 8 * (q >> 2)) * ldy] = acc[i][j][q];
        }
    }
  }
  if (FULL && p.vec_store && p.Y != nullptr) {
    // full tiles, 16-byte aligned rows: the wave's sub-tile goes through its private LDS region and leaves as whole
    // row segments with dwordx4 stores -- 4x fewer store instructions (the epilogue is store-issue bound)
    spg_store_tile_vec<IT / WI, JT / WJ>(acc, red + wave * SPG_EPI_WAVE_FLOATS(IT / WI, JT / WJ),
                                         p.Y + (m0 + roww) * p.ldy + n0 + colw, (unsigned)p.ldy, lane);
  }
  SPG_EP(0);
  // ---- BatchNorm partials: every WAVE writes (mean, M2 = sum (y - mean)^2) of its own rows -- no LDS, no barrier;
  //      spg_bn_finalize_kernel combines the ntile*WI partials with Chan's formula ----
  const int nvw = min(max(mvalid - roww, 0), IT / WI);      // valid rows of this wave
  const long part = (long)tile * WI + wi;
  if (p.stat != nullptr || p.stat_slots != nullptr) {
    const float inv = nvw > 0 ? 1.f / (float)nvw : 0.f;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + colw + 32 * j + r;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (FULL || roww + 32 * i + spg_acc_row(q, h) < mvalid) s += acc[i][j][q];
      s += __shfl_xor(s, 32, 64);
      const float mean = s * inv;
      float m2 = 0.f;
      if constexpr (FULL) {      // (the differences two at a time: v_pk_add_f32 on the accumulator pairs; the chain of FMAs as before)
        const spg_f32x2 mean2 = {mean, mean};
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            const spg_f32x2 d = spg_f32x2{acc[i][j][q], acc[i][j][q + 1]} - mean2;
            m2 = fmaf(d[0], d[0], m2);
            m2 = fmaf(d[1], d[1], m2);
          }
      } else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (roww + 32 * i + spg_acc_row(q, h) < mvalid) {
              const float d = acc[i][j][q] - mean;
              m2 = fmaf(d, d, m2);
            }
      }
      m2 += __shfl_xor(m2, 32, 64);
      if (accum) {                       // persistent stream: merge into the running (rows, mean, M2) of this column
        const float na = sacc.n, nb = (float)nvw, nn = na + nb;
        const float delta = mean - sacc.a[j], f = nn > 0.f ? nb / nn : 0.f;
        sacc.a[j] += delta * f;
        sacc.b[j] += m2 + delta * delta * (na * f);
      } else if (h == 0 && (FULL || col < p.N)) {
        if (p.stat_slots != nullptr) {
          if (nvw > 0) spg_slots_add_fwd(p.stat_slots, p.N, col, (float)nvw, mean, m2);
        } else {
          p.stat[(part * 2 + 0) * p.N + col] = mean;
          p.stat[(part * 2 + 1) * p.N + col] = m2;
        }
      }
    }
    if (accum) sacc.n += (float)nvw;
    else if (lane == 0 && wj == 0 && n0 == 0 && p.stat_slots == nullptr) p.stat_cnt[part] = (float)nvw;
  }
  SPG_EP(1);
  // ---- max-pool over the rows of the tile, fused (see SpgGemmParams): every lane keeps ONE extremum of its column --
  //      the maximum of key = v * sign, sign = -1 where the BatchNorm scale is negative -- with the first row that
  //      attains it; the two lane halves and then the WI row-waves of the workgroup are combined through LDS ----
  if (p.pool_out != nullptr) {
    float* xch = red + 4 * SPG_EPI_WAVE_FLOATS(IT / WI, JT / WJ);        // [WI][JT] keys, [WI][JT] rows: behind the staging regions
    int* xci = reinterpret_cast<int*>(xch + WI * JT);
    float* xsg = xch + 2 * WI * JT;                                      // [JT] signs
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int cl = colw + 32 * j + r, col = n0 + cl;
      const bool colok = FULL || col < p.N;
      // (a persistent stream loaded the sign of its columns once, in front of its tile loop: a global load HERE sits behind the
      //  tile's stores and the next tile's prefetch in the in-order vector-memory counter)
      const float sg = sacc.has_sg ? (sacc.sg[j] < 0.f ? -1.f : 1.f) : ((p.pool_sign != nullptr && p.pool_sign[colok ? col : 0] < 0.f) ? -1.f : 1.f);
      float kb = -FLT_MAX;
      int ib = INT_MAX;
      const spg_f32x2 sg2 = {sg, sg};
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q2 = 0; q2 < 16; q2 += 2) {
          const spg_f32x2 key2 = spg_f32x2{acc[i][j][q2], acc[i][j][q2 + 1]} * sg2;      // (v_pk_mul_f32)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            // rows are visited in increasing order: a strict comparison keeps the FIRST extremum (torch's tie rule)
            const int q = q2 + e;
            const int row = roww + 32 * i + spg_acc_row(q, h);
            const float key = key2[e];
            const bool gt = (FULL || row < mvalid) && key > kb;
            kb = gt ? key : kb; ib = gt ? row : ib;
          }
        }
      const float ok = __shfl_xor(kb, 32, 64);
      const int oi = __shfl_xor(ib, 32, 64);
      if (ok > kb || (ok == kb && oi < ib)) { kb = ok; ib = oi; }
      if (h == 0) { xch[wi * JT + cl] = kb; xci[wi * JT + cl] = ib; if (wi == 0) xsg[cl] = sg; }
    }
    SPG_EP(2);
    __syncthreads();
    for (int cl = tid; cl < JT; cl += SPG_THREADS) {
      const int col = n0 + cl;
      if (!FULL && col >= p.N) continue;
      float kb = xch[cl];
      int ib = xci[cl];
#pragma unroll
      for (int w = 1; w < WI; ++w) {
        const float ok = xch[w * JT + cl];
        const int oi = xci[w * JT + cl];
        if (ok > kb || (ok == kb && oi < ib)) { kb = ok; ib = oi; }
      }
      const float sg = xsg[cl];          // (the sign the row-waves used, through LDS: no second global load)
      p.pool_out[(long)tile * p.pool_ld + col] = kb * sg;
      if (p.pool_idx != nullptr) p.pool_idx[(long)tile * p.pool_ld + col] = ib;
    }
    if (n0 == 0 && p.pool_extra != nullptr)
      for (int e = tid; e < p.pool_nextra; e += SPG_THREADS)
        p.pool_out[(long)tile * p.pool_ld + p.N + e] = p.pool_extra[(long)tile * p.pool_nextra + e];
    // (no barrier here any more: every caller follows the epilogue with a workgroup barrier before anything writes this LDS region
    //  again -- the tile loop's own barrier, the one in front of the stream's final statistics merge -- or with nothing at all)
    SPG_EP(3);
  }
}

// per-wave backward sums (channel quads over the lanes, LPR lanes per row) -> one partial row pair
template <int IT, int JT, int WI, int WJ>
__device__ __forceinline__ void spg_bwd_stat_store(const SpgGemmParams& p, f32x4 s1, f32x4 s2, long part, int n0) {
  constexpr int CW = JT / WJ, LPR = CW / 4;
  const int tid = threadIdx.x + spg_opaque_zero(), lane = tid & 63, wave = tid >> 6;
  const int wj = wave % WJ;
  const int col = n0 + wj * CW + 4 * (lane % LPR);
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
  }
  if (lane < LPR) {
    if (p.stat_slots != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < p.n_mask) spg_slots_add(p.stat_slots, p.n_mask, col + e, (double)s1[e], (double)s2[e]);
    } else {
      *reinterpret_cast<f32x4*>(p.stat + (part * 2 + 0) * p.N + col) = s1;
      *reinterpret_cast<f32x4*>(p.stat + (part * 2 + 1) * p.N + col) = s2;
    }
  }
}

// Vector form of the backward epilogue for full tiles with 16-byte aligned rows: the wave's accumulators go through its
// private LDS region; every lane then owns 4 consecutive channels of a few rows -- one dwordx4 load of the producer's
// output, the ReLU mask, one dwordx4 store, and the two BatchNorm-backward sums in registers (reduced over the 4 lane
// groups that share a channel quad).  16 + 16 memory instructions per lane instead of 64 + 64.
template <int IT, int JT, int WI, int WJ>
__device__ __forceinline__ void spg_epilogue_bwd_vec(const SpgGemmParams& p, f32x16 (&acc)[IT / WI / 32][JT / WJ / 32],
                                                     float* __restrict__ red, int tile, long m0, int n0,
                                                     SpgStatAcc<JT / WJ / 32>& sacc, const bool accum) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32, RW = IT / WI, CW = JT / WJ;
  constexpr int LD = CW + 8, LPR = CW / 4, RPI = 64 / LPR, NIT = SPG_EPI_PIECE_ROWS / RPI;
  static_assert(SPG_EPI_PIECE_ROWS % RPI == 0, "a piece is a whole number of store instructions");
  const int tid = threadIdx.x + spg_opaque_zero(), lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  const int colw = wj * CW, roww = wi * RW;
  float* st = red + wave * SPG_EPI_WAVE_FLOATS(RW, CW);
  const bool do_stats = (p.stat != nullptr || p.stat_slots != nullptr) && p.mmean != nullptr;           // uniform
  const bool do_mask = p.mask_relu != 0 && p.Yp != nullptr;                // uniform
  const bool use_y = p.Yp != nullptr && (do_stats || do_mask);
  const int lr = lane / LPR, lc = 4 * (lane % LPR);
  const int col = n0 + colw + lc;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, mean = sh, rstd = sh;
  if (p.ms != nullptr) { sc = *reinterpret_cast<const f32x4*>(p.ms + col); sh = *reinterpret_cast<const f32x4*>(p.mt + col); }
  if (do_stats) { mean = *reinterpret_cast<const f32x4*>(p.mmean + col); rstd = *reinterpret_cast<const f32x4*>(p.mrstd + col); }
  const float* yp = p.Yp + (m0 + roww) * p.ldyp + n0 + colw;
  float* yo = p.Y + (m0 + roww) * p.ldy + n0 + colw;
  const unsigned op0 = (unsigned)lr * (unsigned)p.ldyp + (unsigned)lc, oo0 = (unsigned)lr * (unsigned)p.ldy + (unsigned)lc;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int rb = 32 * i + 16 * half;       // first row of this piece inside the wave's sub-tile
      f32x4 yv[NIT];
      if (use_y) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) yv[u] = *reinterpret_cast<const f32x4*>(yp + op0 + (unsigned)(rb + RPI * u) * (unsigned)p.ldyp);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) st[((u & 3) + 8 * (u >> 2) + 4 * h) * LD + 32 * j + r] = acc[i][j][8 * half + u];
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        f32x4 v = *reinterpret_cast<const f32x4*>(st + (lr + RPI * u) * LD + lc);
        const f32x4 y = use_y ? yv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (do_mask) {
          const f32x4 act = spg_fma4(y, sc, sh);                                  // (packed: two channels per VALU operation)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(act[e] > 0.f)) v[e] = 0.f;
        }
        if (do_stats) {
          s1 += v;
          s2 = spg_fma4(v, (y - mean) * rstd, s2);
        }
        *reinterpret_cast<f32x4*>(yo + oo0 + (unsigned)(rb + RPI * u) * (unsigned)p.ldy) = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  if (accum) {                         // persistent stream: running sums, written once when the stream ends
    sacc.s1 += s1; sacc.s2 += s2;
  } else if (p.stat != nullptr || p.stat_slots != nullptr) {
    spg_bwd_stat_store<IT, JT, WI, WJ>(p, s1, s2, (long)tile * WI + wi, n0);
  }
}

// backward: ReLU mask of the producer layer, store dz, BatchNorm-backward partial sums (sum dz, sum dz*xhat)
template <int IT, int JT, int WI, int WJ, bool FULL>
__device__ __forceinline__ void spg_epilogue_bwd(const SpgGemmParams& p, f32x16 (&acc)[IT / WI / 32][JT / WJ / 32],
                                                 float* __restrict__ red, int tile, long m0, int mvalid, int n0) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32;
  const int tid = threadIdx.x + spg_opaque_zero(), lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  const int colw = wj * (JT / WJ), roww = wi * (IT / WI);
  const bool do_stats = (p.stat != nullptr || p.stat_slots != nullptr) && p.mmean != nullptr;           // uniform
  const bool do_mask = p.mask_relu != 0 && p.Yp != nullptr;                // uniform
  const bool use_y = p.Yp != nullptr && (do_stats || do_mask);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + colw + 32 * j + r;
    const bool colok = FULL || col < p.N;
    const bool inm = FULL || col < p.n_mask;                               // channel of the producer layer
    const int cc = (colok && inm) ? col : 0;
    float sc = 1.f, sh = 0.f, mean = 0.f, rstd = 0.f;
    if (p.ms != nullptr) { sc = p.ms[cc]; sh = p.mt[cc]; }
    if (do_stats) { mean = p.mmean[cc]; rstd = p.mrstd[cc]; }
    // wave-uniform bases (SGPRs) + 32-bit lane offsets: one VGPR per address instead of a 64-bit pair
    const float* yp = p.Yp + m0 * p.ldyp + n0;
    float* yo = p.Y + m0 * p.ldy + n0;
    const unsigned ldyp = (unsigned)p.ldyp, ldyo = (unsigned)p.ldy;
    const unsigned cl = (unsigned)(colw + 32 * j + r);
    const unsigned op0 = (unsigned)(roww + 4 * h) * ldyp + (colok ? cl : 0u);
    const unsigned oo0 = (unsigned)(roww + 4 * h) * ldyo + cl;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      float yv[16];
      if (use_y) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 32 * i + spg_acc_row(q, h);
          const bool ok = FULL || (colok && roww + row < mvalid);
          yv[q] = yp[ok ? op0 + (unsigned)(32 * i + (q & 3) + 8 * (q >> 2)) * ldyp : 0u];   // unconditional (clamped) load
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * i + spg_acc_row(q, h);
        const bool ok = FULL || (colok && roww + row < mvalid);
        float v = acc[i][j][q];
        const float y = use_y ? yv[q] : 0.f;
        if (do_mask && inm && !(fmaf(y, sc, sh) > 0.f)) v = 0.f;
        if (FULL || ok) yo[oo0 + (unsigned)(32 * i + (q & 3) + 8 * (q >> 2)) * ldyo] = v;
        if (do_stats) {
          const float w = (ok && inm) ? v : 0.f;
          s1 += w;
          s2 = fmaf(w, (y - mean) * rstd, s2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the 16-load batches of different sub-tiles apart (register pressure)
    }
    if (p.stat != nullptr || p.stat_slots != nullptr) {     // per-wave partial sums, no LDS / barrier
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0 && (FULL || col < p.N)) {
        const long part = (long)tile * WI + wi;
        if (p.stat_slots != nullptr) {
          if (col < p.n_mask) spg_slots_add(p.stat_slots, p.n_mask, col, (double)s1, (double)s2);
        } else {
          p.stat[(part * 2 + 0) * p.N + col] = s1;
          p.stat[(part * 2 + 1) * p.N + col] = s2;
        }
      }
    }
  }
}

// epilogue of one finished tile (all variants): `red` = LDS staging region that nobody reads any more
// accum (persistent streams: whole tiles, vector stores -- host): statistics are accumulated in `sacc` instead of written
template <int IT, int JT, int WI, int WJ, bool WRED, bool FULL, bool BIAS_DONE>
__device__ __forceinline__ void spg_tile_epilogue(const SpgGemmParams& p, f32x16 (&acc)[IT / WI / 32][JT / WJ / 32],
                                                  float* __restrict__ red, int tile, long m0, int mvalid, int n0,
                                                  SpgStatAcc<JT / WJ / 32>& sacc, const bool accum, unsigned long long* edbg = nullptr) {
  if constexpr (!WRED) {      // forward kernels: weights [N,K]; backward (dgrad) kernels: untransposed weights [K,N]
    if (mvalid == IT && n0 + JT <= p.N) spg_epilogue_fwd<IT, JT, WI, WJ, true, BIAS_DONE>(p, acc, red, tile, m0, mvalid, n0, sacc, accum, edbg);
    else spg_epilogue_fwd<IT, JT, WI, WJ, false, BIAS_DONE>(p, acc, red, tile, m0, mvalid, n0, sacc, false);
  } else {
    const bool full = mvalid == IT && n0 + JT <= p.N && (n0 + JT <= p.n_mask || (p.stat == nullptr && p.stat_slots == nullptr && !p.mask_relu));
    if (FULL && full && p.vec_store) spg_epilogue_bwd_vec<IT, JT, WI, WJ>(p, acc, red, tile, m0, n0, sacc, accum);
    else if (full) spg_epilogue_bwd<IT, JT, WI, WJ, true>(p, acc, red, tile, m0, mvalid, n0);
    else spg_epilogue_bwd<IT, JT, WI, WJ, false>(p, acc, red, tile, m0, mvalid, n0);
  }
}

// AMODE >= 0: operand mode of A known at compile time, vector + software-pipelined main loop (host guarantees the
// alignment conditions); AMODE < 0: generic scalar staging (channel-major clouds, unaligned leading dimensions).
// FULL: every tile of the launch is complete (rows, output channels, reduction chunks): the masked staging pipes are
// replaced by the loop-invariant-address fast pipes (spg_common.h), and the kernel is a PERSISTENT CHUNK STREAM: a
// workgroup takes row tiles tile, tile + rstride, ... of its column tile and treats their reduction chunks as one
// stream -- the loads of the next tile's first two chunks are issued under the last two chunks of the current one and
// its first chunk is already in LDS when the epilogue starts, so neither workgroup launch latency nor first-load
// latency is paid per tile (measured before: wave slots empty 20 % of the kernel, conv5 at 0.69 of the MFMA peak even
// without any epilogue).  A launch with one tile per workgroup (rstride >= ntile) is the degenerate case.
// STREAM: compiled with the multi-tile stream (persistent launches); without it has_next is a compile-time false
// PREC: 0 = fp32 MFMA; 1 / 3 = bf16 / split-bf16 MFMA (FULL only; spg_common.h): operands are converted while staging,
// the weights come pre-split (p.Wb, out-major for both the forward and -- pre-transposed -- the data gradient)
// (bx, by): the workgroup's position in the launch grid -- blockIdx for a stand-alone launch, a virtual position inside its
// job for a workgroup of a grouped launch (spg_multi_kernel below)
#ifdef SPG_ATTRIBUTION
// (attribution builds only; tools/fwd_phase_timing.py) shader cycles of wave 0 of every workgroup of the persistent forward streams,
// by phase: [class][0 entry -> first chunk loop, 1 chunk loops, 2 epilogues, 3 last epilogue -> exit, 4 workgroups, 5 tiles];
// class = 0: K 64, 1: K 128 / N 128, 2: N 256 (pooled layer), 3: K 64 with pooling (STN)
__device__ unsigned long long spg_fwd_phase_t[4][14];
extern "C" int spg_fwd_phase_times(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spg_fwd_phase_t), sizeof(unsigned long long) * 56) != hipSuccess) return -1;
  if (clear) { unsigned long long z[56] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(spg_fwd_phase_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define SPG_F0() unsigned long long ft__ = __builtin_readcyclecounter(), fp__[4] = {0, 0, 0, 0}, fn__ = 0, ep__[4] = {0, 0, 0, 0}, en__[4] = {0, 0, 0, 0}, et0__ = ft__
#define SPG_FP(k) { const unsigned long long n__ = __builtin_readcyclecounter(); fp__[k] += n__ - ft__; ft__ = n__; }
#define SPG_FTILE() ++fn__
#define SPG_FN(k) { const unsigned long long n__ = __builtin_readcyclecounter(); en__[k] += n__ - et0__; et0__ = n__; }
#define SPG_FN_FIRST(k) if (fn__ == 0) SPG_FN(k)
#define SPG_FEND() if (STREAM && !WRED && threadIdx.x == 0) { const int c__ = p.N > 128 ? 2 : (p.K > 64 ? 1 : (p.pool_out != nullptr ? 3 : 0)); \
    for (int k__ = 0; k__ < 4; ++k__) atomicAdd(&spg_fwd_phase_t[c__][k__], fp__[k__]); atomicAdd(&spg_fwd_phase_t[c__][4], 1ull); atomicAdd(&spg_fwd_phase_t[c__][5], fn__); for (int k__ = 0; k__ < 4; ++k__) { atomicAdd(&spg_fwd_phase_t[c__][6 + k__], ep__[k__]); atomicAdd(&spg_fwd_phase_t[c__][10 + k__], en__[k__]); } }
#else
#define SPG_F0()
#define SPG_FP(k)
#define SPG_FTILE()
#define SPG_FN(k)
#define SPG_FN_FIRST(k)
#define SPG_FEND()
#endif

template <int IT, int JT, int WI, int WJ, bool WRED, int AMODE, bool FULL = false, bool STREAM = false, int PREC = 0>
__device__ __forceinline__ void spg_rowgemm_body(const SpgGemmParams& p, const int bx, const int by) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32;
  static_assert(WI * WJ == 4 && TI >= 1 && TJ >= 1, "4 waves per workgroup");
  extern __shared__ f32x4 smem[];
  f32x4* As = smem;                                   // buffer b: A at smem + b*(A_F4+B_F4), weights right behind it
  f32x4* Bs = smem + (SPG_KC / 4) * (IT + 1);
  float* Bsr = reinterpret_cast<float*>(Bs);

  SPG_F0();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  constexpr int A_F4 = (SPG_KC / 4) * (IT + 1);                                  // float4 slots of one A buffer
  constexpr int B_F4 = (WRED && PREC == 0) ? SPG_KC * (JT + 4) / 4 : (SPG_KC / 4) * (JT + 1);   // float4 slots of one weight buffer
  f32x16 acc[TI][TJ];

  // BatchNorm of the layer that produced operand `a`: its statistics arrive as fixed-point slots and are finished here (every
  // workgroup; spg_gemm.h) -- behind the barrier at its end the scale / shift arrays the staging pipes read exist
  // the rows behind this launch's statistics travel with them (spg_fold.h): one workgroup counts them
  if (p.stat_slots != nullptr && bx == 0 && by == 0 && threadIdx.x == 0)
    spg_slots_count_add(p.stat_slots, WRED ? p.n_mask : p.N, p.stat_rows != 0 ? p.stat_rows : (long)p.M);
  // (round 5: on the full-tile path the fold runs BEHIND the issue of the first chunk's raw global loads -- they do not depend on
  //  the constants, only the staging arithmetic does -- so one global round trip of every folding launch hides behind the other)
  auto fold_now = [&]() __attribute__((always_inline)) {
    if constexpr (!WRED) {
      if (p.fold.slots != nullptr) spg_bn_fold_fwd(p.fold, bx == 0 && by == 0);
    } else {
      // data gradient launched NEXT TO the layer's weight gradient (grouped launch): it finishes the BatchNorm-backward
      // constants its own staging reads itself -- the weight gradient's workgroups (which write the same bits, and dgamma /
      // dbeta) are not ordered before it any more
      if (p.fold_bwd.slots != nullptr) spg_bn_fold_bwd(p.fold_bwd, false);
    }
  };
  if constexpr (!(AMODE >= 0 && FULL)) fold_now();

  if constexpr (AMODE >= 0 && FULL) {
    static_assert(4 * SPG_EPI_WAVE_FLOATS(IT / WI, JT / WJ) + 2 * WI * JT + JT <= 4 * (A_F4 + B_F4), "epilogue staging + pooling exchange must fit one LDS buffer");
    int tile = bx, ct = by;
    if (p.remap) {
      // XCD-aware item map (workgroup b runs on XCD b % 8): the column tiles of one row tile are consecutive workgroups
      // of ONE XCD (the second reader of an A tile finds it in that XCD's L2); a workgroup keeps its column tile
      const int lin = bx, j = lin >> 3;
      ct = j % p.ncol;
      tile = (j / p.ncol) * 8 + (lin & 7);
    }
    const int ntile = p.ntile, rstride = p.rstride;
    if (tile >= ntile) return;
    const int n0 = ct * JT;
    long m0 = (long)tile * p.rows_per_tile;
    int mvalid = (int)min((long)p.rows_per_tile, (long)p.M - m0);     // == IT for every tile of a multi-tile stream (host)
    constexpr bool BIAS_IN_ACC = !WRED;
    SpgStatAcc<TJ> sacc;                         // STREAM: statistics of this workgroup's tiles, one partial at the end
    sacc.n = 0.f;
#pragma unroll
    for (int j = 0; j < TJ; ++j) { sacc.a[j] = 0.f; sacc.b[j] = 0.f; }
    sacc.s1 = f32x4{0.f, 0.f, 0.f, 0.f}; sacc.s2 = sacc.s1;
    // loop-invariant per-lane values of the workgroup's column tile, loaded ONCE and FIRST (in front of the tile's loads: their latency
    // hides behind the BatchNorm fold): the bias the accumulators start from and (pooled layer) the sign of the BatchNorm scale -- a
    // global load inside the tile loop waits, through the in-order vector-memory counter, for the previous tile's stores and the
    // prefetched loads
    float bias_v[TJ];          // (raw loaded values: nothing below may USE them before the first chunk is on its way -- a use is a wait)
    sacc.has_sg = !WRED && p.pool_out != nullptr;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wj * (JT / WJ) + 32 * j + r;
      bias_v[j] = 0.f; sacc.sg[j] = 1.f;
      if (BIAS_IN_ACC && p.bias != nullptr) bias_v[j] = p.bias[col < p.N ? col : 0];      // uniform
      if (sacc.has_sg && p.pool_sign != nullptr) sacc.sg[j] = p.pool_sign[col < p.N ? col : 0];
    }
    // two register sets: iteration c issues the global loads of chunk c+2 into one set (slots 0-2), then finishes chunk
    // c+1 from the other set (loaded during iteration c-1, i.e. a full chunk of MFMAs ago) into the idle LDS buffer
    SpgRowsFast<AMODE, IT> pa0, pa1;
    SpgWeightFast<JT> pw0, pw1;
    SpgWeightRedFast<JT> pwr0, pwr1;
    SpgWeightBf16<JT, PREC ? PREC : 1> pb0, pb1;
    constexpr int NIA = SpgRowsFast<AMODE, IT>::NI;
    constexpr int NIW = PREC ? SpgWeightBf16<JT, PREC ? PREC : 1>::NI : (WRED ? SpgWeightRedFast<JT>::NI : SpgWeightFast<JT>::NI);
    // one place per weight-pipe operation: the three layouts (out-major fp32, red-major fp32, out-major bf16) are
    // compile-time alternatives
    auto w_load = [&](SpgWeightFast<JT>& pw, SpgWeightRedFast<JT>& pwr, SpgWeightBf16<JT, PREC ? PREC : 1>& pb, int k, int i) __attribute__((always_inline)) {
      if constexpr (PREC != 0) pb.load_part(p.Wb, p.ldwb, n0, k, i);
      else if constexpr (WRED) pwr.load_part(p.W, p.ldw, n0, k, i);
      else pw.load_part(p.W, p.ldw, n0, k, i);
    };
    auto w_store = [&](const SpgWeightFast<JT>& pw, const SpgWeightRedFast<JT>& pwr, const SpgWeightBf16<JT, PREC ? PREC : 1>& pb, f32x4* B, int i) __attribute__((always_inline)) {
      if constexpr (PREC != 0) pb.store_part(B, i);
      else if constexpr (WRED) pwr.store_part(reinterpret_cast<float*>(B), i);
      else pw.store_part(B, i);
    };
    auto a_store = [&](const SpgRowsFast<AMODE, IT>& pa, f32x4* A, int i) __attribute__((always_inline)) {
      if constexpr (PREC != 0) pa.template store_part_bf16<PREC ? PREC : 1>(A, i);
      else pa.store_part(A, i);
    };
    static_assert(NIA + NIW + 4 <= SPG_KC / 2, "staging pieces must fit the MFMA slots of a chunk");
    const int nchunk = p.K / SPG_KC;
    pa0.init(p.a, mvalid); pa1.init(p.a, mvalid);
    if constexpr (PREC != 0) { pb0.init(p.ldwb, n0, p.N, p.wb_part_bytes); pb1.init(p.ldwb, n0, p.N, p.wb_part_bytes); }
    else if (WRED) { pwr0.init(p.ldw, n0, p.N); pwr1.init(p.ldw, n0, p.N); } else { pw0.init(p.ldw, n0, p.N); pw1.init(p.ldw, n0, p.N); }
    // chunk 0 -> LDS buffer 0; chunk 1 in flight in set 1
    {
#pragma unroll
      for (int i = 0; i < NIA; ++i) pa0.load_part(p.a, m0, 0, i);
#pragma unroll
      for (int i = 0; i < NIW; ++i) w_load(pw0, pwr0, pb0, 0, i);
      SPG_FN(0);
      fold_now();                      // (ends with a workgroup barrier: the constants `prepare` reads exist behind it)
      SPG_FN(1);
      pa0.prepare(p.a, tile, 0);
#pragma unroll
      for (int i = 0; i < NIA; ++i) a_store(pa0, As, i);
#pragma unroll
      for (int i = 0; i < NIW; ++i) w_store(pw0, pwr0, pb0, Bs, i);
      const int k1 = nchunk > 1 ? SPG_KC : 0;
      pa1.prepare(p.a, tile, k1);
#pragma unroll
      for (int i = 0; i < NIA; ++i) pa1.load_part(p.a, m0, k1, i);
#pragma unroll
      for (int i = 0; i < NIW; ++i) w_load(pw1, pwr1, pb1, k1, i);
    }
    __syncthreads();
    // (the loop-invariant loads of the top become USABLE here: the compiler had moved the sign's compare right behind its load -- a
    //  wait for a global round trip in front of the first tile load)
#pragma unroll
    for (int j = 0; j < TJ; ++j) asm volatile("" : "+v"(sacc.sg[j]), "+v"(bias_v[j]));
    SPG_FN(2);
    const int tile0 = tile;                      // < rstride: index of this workgroup among those of its column tile
    const bool accum = STREAM && (p.stat != nullptr || p.stat_slots != nullptr) && p.stat_accum;
    // backward kernels (two operand streams + four constant arrays per register set): the loads of the next tile's SECOND
    // chunk are issued after the epilogue instead of under the last chunk -- nothing but the accumulators and the
    // loop-invariant offsets is then live across the epilogue (no spills)
    constexpr bool DEFER2 = WRED;      // (tried for the forward as well, round 6: 1.139 -> 1.143 ms per step)
    // (a 2 us start offset for the second round of workgroups -- anti-phase of the two workgroups of a CU -- gained 0.4 % while the
    //  streams' statistics still lived in scratch memory; with them in registers it measures 0.0 % at 1 and 8 scenes: removed)
    for (;;) {
      tile = __builtin_amdgcn_readfirstlane(tile);             // wave-uniform by construction: keep the tile stream in SGPRs
      m0 = (long)tile * p.rows_per_tile;
      const int nxt = tile + rstride;
      const bool has_next = STREAM && nxt < ntile;               // uniform
      const long m0n = has_next ? (long)nxt * p.rows_per_tile : m0;
      const int tilen = has_next ? nxt : tile;
      // accumulators start from the bias (forward) / zero
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float bv = n0 + wj * (JT / WJ) + 32 * j + r < p.N ? bias_v[j] : 0.f;
        asm volatile("" : "+v"(bv));      // (opaque per tile: the 16-register splat below must not be hoisted out of the tile loop -- it was, and spilled)
        // (as 64-bit values: v_mov_b64, 8 instead of 16 VALU operations per 32 x 32 block)
        typedef double spg_f64x8 __attribute__((ext_vector_type(8)));
        const double bd = __builtin_bit_cast(double, spg_f32x2{bv, bv});
        const spg_f64x8 b8 = {bd, bd, bd, bd, bd, bd, bd, bd};
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[i][j] = __builtin_bit_cast(f32x16, b8);
      }
      // body(c, F, L): MFMAs of chunk c; finish chunk c+1 from set F; load chunk c+2 into set L (the one finished last
      // time).  Chunks nchunk, nchunk+1 are chunks 0, 1 of the NEXT tile of this workgroup (clamped re-loads of the last
      // chunk when there is none: harmless).
      auto body = [&](int c, int k2, long ml, int tl, bool noload, SpgRowsFast<AMODE, IT>& paF, SpgWeightFast<JT>& pwF, SpgWeightRedFast<JT>& pwrF,
                      SpgWeightBf16<JT, PREC ? PREC : 1>& pbF, SpgRowsFast<AMODE, IT>& paL, SpgWeightFast<JT>& pwL, SpgWeightRedFast<JT>& pwrL,
                      SpgWeightBf16<JT, PREC ? PREC : 1>& pbL) __attribute__((always_inline)) {
        const int buf = c & 1;
        const f32x4* Ac = As + buf * (A_F4 + B_F4);
        const f32x4* Bc = Bs + buf * (A_F4 + B_F4);
        f32x4* An = As + (buf ^ 1) * (A_F4 + B_F4);
        f32x4* Bn = Bs + (buf ^ 1) * (A_F4 + B_F4);
        auto piece = [&](int slot) __attribute__((always_inline)) {
          if (slot == 0) {
            if (!(DEFER2 && noload)) {
              paL.prepare(p.a, tl, k2);
#pragma unroll
              for (int i = 0; i < (NIA + 1) / 2; ++i) paL.load_part(p.a, ml, k2, i);
            }
          } else if (slot == 1) {
            if (!(DEFER2 && noload)) {
#pragma unroll
              for (int i = (NIA + 1) / 2; i < NIA; ++i) paL.load_part(p.a, ml, k2, i);
            }
          } else if (slot == 2) {
            if (!(DEFER2 && noload)) {
#pragma unroll
              for (int i = 0; i < NIW; ++i) w_load(pwL, pwrL, pbL, k2, i);
            }
          } else if (slot < 3 + NIA) {
            a_store(paF, An, slot - 3);
          } else if (slot < 3 + NIA + NIW) {
            w_store(pwF, pwrF, pbF, Bn, slot - 3 - NIA);
          }
        };
        if constexpr (PREC != 0) {
          int sa[TI], sb[TJ];
#pragma unroll
          for (int i = 0; i < TI; ++i) sa[i] = wi * (IT / WI) + r + 32 * i;
#pragma unroll
          for (int j = 0; j < TJ; ++j) sb[j] = wj * (JT / WJ) + r + 32 * j;
          spg_mfma_chunk_bf16_il<TI, TJ, PREC>(Ac, Bc, IT + 1, JT + 1, sa, sb, h, acc, piece);
        }
        else if (WRED) spg_mfma_chunk_or_il<TI, TJ>(Ac, reinterpret_cast<const float*>(Bc), IT + 1, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
        else spg_mfma_chunk_il<TI, TJ>(Ac, Bc, IT + 1, JT + 1, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
        if (!(STREAM && (SPG_DBG(p) & 32))) __syncthreads();      // (attribution switch: main loop without barriers)
      };
      const int reps = STREAM ? 1 + ((SPG_DBG(p) >> 8) & 15) : 1;      // (attribution switch: the chunk loop repeated -> steady-state rate)
      SPG_FN_FIRST(3);
      SPG_FP(fn__ == 0 ? 0 : 2);
      for (int rep = 0; rep < reps; ++rep)
      for (int c = 0; c < nchunk; c += 2) {
        // where the loads of chunk c+2 / c+3 come from: this tile, the next tile of the stream, or (at the very end) a
        // clamped re-load of the last chunk -- plain scalar values, computed outside the lambdas
        int ka = c + 2, kb = c + 3;
        long ma = m0, mb = m0;
        if (STREAM && (SPG_DBG(p) & 16)) ma = mb = (long)(tile & 15) * p.rows_per_tile;     // (attribution switch: A from L2)
        int ta = tile, tb = tile;
        if (ka >= nchunk) { if (has_next) { ka -= nchunk; ma = (STREAM && (SPG_DBG(p) & 16)) ? ma : m0n; ta = tilen; } else ka = nchunk - 1; }
        if (kb >= nchunk) { if (has_next) { kb -= nchunk; mb = (STREAM && (SPG_DBG(p) & 16)) ? mb : m0n; tb = tilen; } else kb = nchunk - 1; }
        body(c, ka * SPG_KC, ma, ta, false, pa1, pw1, pwr1, pb1, pa0, pw0, pwr0, pb0);          // set 1 holds chunk c+1; set 0 is free for chunk c+2
        if (c + 1 < nchunk) body(c + 1, kb * SPG_KC, mb, tb, has_next && c + 2 >= nchunk, pa0, pw0, pwr0, pb0, pa1, pw1, pwr1, pb1);
      }
      // Here (nchunk even when there is a next tile -- host): LDS buffer 0 holds chunk 0 of the next tile, set 1 its chunk
      // 1 (in flight); buffer 1 was read by the last chunk and is free: the epilogue stages through it.
      float* red = reinterpret_cast<float*>(smem + (A_F4 + B_F4));
      SPG_FP(1); SPG_FTILE();
      if (STREAM && SPG_DBG(p)) {       // timing attribution only (spg_tune key 3; results are WRONG): parts of the epilogue off
        if (SPG_DBG(p) & 8) {
          if (acc[0][0][0] + acc[TI - 1][TJ - 1][5] == 123.456f) p.Y[0] = 0.f;
        } else {
          SpgGemmParams q = p;
          if (SPG_DBG(p) & 1) q.Y = nullptr;
          if (SPG_DBG(p) & 2) q.stat = nullptr;
          if (SPG_DBG(p) & 4) q.pool_out = nullptr;
          spg_tile_epilogue<IT, JT, WI, WJ, WRED, true, BIAS_IN_ACC>(q, acc, red, tile, m0, mvalid, n0, sacc, false);
        }
      } else
#ifdef SPG_ATTRIBUTION
      spg_tile_epilogue<IT, JT, WI, WJ, WRED, true, BIAS_IN_ACC>(p, acc, red, tile, m0, mvalid, n0, sacc, accum, ep__);
#else
      spg_tile_epilogue<IT, JT, WI, WJ, WRED, true, BIAS_IN_ACC>(p, acc, red, tile, m0, mvalid, n0, sacc, accum);
#endif
      if (!has_next) break;
      tile = nxt; m0 = m0n;
      if (DEFER2) {
        pa1.prepare(p.a, tile, SPG_KC);
#pragma unroll
        for (int i = 0; i < NIA; ++i) pa1.load_part(p.a, m0, SPG_KC, i);
#pragma unroll
        for (int i = 0; i < NIW; ++i) w_load(pw1, pwr1, pb1, SPG_KC, i);
      }
      __syncthreads();          // the staging region is overwritten by the next chunk's finish stage
    }
    SPG_FP(2);
    if (accum) {
      // The workgroup's ONE statistics partial: the WI row-waves hold (rows, mean, M2) / (sum, sum) of the same columns;
      // waves 1.. hand theirs to wave 0 through LDS (free now), which merges in fixed order and stores -- WI times fewer
      // partials for the finalize kernel (<= 512 per layer: no sliced reduction, no last-arrival ticket).
      float* xch = reinterpret_cast<float*>(smem);             // [WI][JT][2]
      const long part = tile0;
      __syncthreads();
      if constexpr (!WRED) {
        if (wi != 0 && h == 0) {
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const int cl = wj * (JT / WJ) + 32 * j + r;
            xch[(wi * JT + cl) * 2 + 0] = sacc.a[j];
            xch[(wi * JT + cl) * 2 + 1] = sacc.b[j];
          }
        }
        __syncthreads();
        if (wi == 0 && h == 0) {
          const float nb = sacc.n;                             // every row-wave saw the same number of rows
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const int cl = wj * (JT / WJ) + 32 * j + r, col = n0 + cl;
            float na = sacc.n, mean = sacc.a[j], m2 = sacc.b[j];
#pragma unroll
            for (int w = 1; w < WI; ++w) {
              const float nn = na + nb, f = nn > 0.f ? nb / nn : 0.f;
              const float delta = xch[(w * JT + cl) * 2 + 0] - mean;
              mean += delta * f;
              m2 += xch[(w * JT + cl) * 2 + 1] + delta * delta * (na * f);
              na = nn;
            }
            if (col < p.N) {
              if (p.stat_slots != nullptr) {
                spg_slots_add_fwd(p.stat_slots, p.N, col, na, mean, m2);
              } else {
                p.stat[(part * 2 + 0) * p.N + col] = mean;
                p.stat[(part * 2 + 1) * p.N + col] = m2;
              }
            }
          }
          if (lane == 0 && wj == 0 && n0 == 0 && p.stat_slots == nullptr) p.stat_cnt[part] = sacc.n * (float)WI;
        }
      } else {
        constexpr int CW = JT / WJ, LPR = CW / 4;
        f32x4 s1 = sacc.s1, s2 = sacc.s2;
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
        }
        const int cl = wj * CW + 4 * (lane % LPR);
        if (wi != 0 && lane < LPR) {
          *reinterpret_cast<f32x4*>(xch + (wi * JT + cl) * 2) = s1;
          *reinterpret_cast<f32x4*>(xch + (wi * JT + cl) * 2 + 4) = s2;
        }
        __syncthreads();
        if (wi == 0 && lane < LPR) {
#pragma unroll
          for (int w = 1; w < WI; ++w) {
            s1 += *reinterpret_cast<const f32x4*>(xch + (w * JT + cl) * 2);
            s2 += *reinterpret_cast<const f32x4*>(xch + (w * JT + cl) * 2 + 4);
          }
          if (p.stat_slots != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + cl + e < p.n_mask) spg_slots_add(p.stat_slots, p.n_mask, n0 + cl + e, (double)s1[e], (double)s2[e]);
          } else {
            *reinterpret_cast<f32x4*>(p.stat + (part * 2 + 0) * p.N + n0 + cl) = s1;
            *reinterpret_cast<f32x4*>(p.stat + (part * 2 + 1) * p.N + n0 + cl) = s2;
          }
        }
      }
    }
    SPG_FP(3); SPG_FEND();
    return;
  }

  const int tile = bx;
  const long m0 = (long)tile * p.rows_per_tile;
  const int mvalid = (int)min((long)p.rows_per_tile, (long)p.M - m0);
  const int n0 = by * JT;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  if constexpr (AMODE >= 0) {
    // software-pipelined main loop, two LDS buffers, ONE barrier per chunk.  Iteration c runs the MFMAs of chunk c and,
    // in their shadow (spg_mfma_chunk_il), finishes chunk c+1 (prologue arithmetic + LDS write into the other buffer;
    // its global loads were issued one iteration earlier) and then issues the global loads of chunk c+2.  The loop is
    // branch-free: past the end, the chunk index is clamped (a redundant, harmless re-load / re-write of the last chunk
    // into the buffer nobody reads any more).
    SpgRowsPipe<AMODE, IT> pa;
    SpgWeightPipe<JT> pw;
    SpgWeightRedPipe<JT> pwr;
    constexpr int NIA = SpgRowsPipe<AMODE, IT>::NI;
    constexpr int NIW = WRED ? SpgWeightRedPipe<JT>::NI : SpgWeightPipe<JT>::NI;
    static_assert(NIA + NIW + 4 <= SPG_KC / 2, "staging pieces must fit the MFMA slots of a chunk");
    const int nchunk = (p.K + SPG_KC - 1) / SPG_KC;
    pa.load(p.a, m0, mvalid, 0, p.K);
    if (WRED) pwr.load(p.W, p.ldw, n0, p.N, 0, p.K); else pw.load(p.W, p.ldw, n0, p.N, 0, p.K);
    pa.store(As);
    if (WRED) pwr.store(Bsr); else pw.store(Bs);
    {
      const int k1 = nchunk > 1 ? SPG_KC : 0;
      pa.load(p.a, m0, mvalid, k1, p.K);
      if (WRED) pwr.load(p.W, p.ldw, n0, p.N, k1, p.K); else pw.load(p.W, p.ldw, n0, p.N, k1, p.K);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      const int k2 = (c + 2 < nchunk ? c + 2 : nchunk - 1) * SPG_KC;
      const int buf = c & 1;
      const f32x4* Ac = As + buf * (A_F4 + B_F4);
      const f32x4* Bc = Bs + buf * (A_F4 + B_F4);
      f32x4* An = As + (buf ^ 1) * (A_F4 + B_F4);
      f32x4* Bn = Bs + (buf ^ 1) * (A_F4 + B_F4);
      auto piece = [&](int slot) __attribute__((always_inline)) {
        if (slot < NIA) {
          pa.store_part(An, slot);
        } else if (slot < NIA + NIW) {
          if (WRED) pwr.store_part(reinterpret_cast<float*>(Bn), slot - NIA); else pw.store_part(Bn, slot - NIA);
        } else if (slot == NIA + NIW) {
          pa.prepare(p.a, k2, p.K);
#pragma unroll
          for (int i = 0; i < (NIA + 1) / 2; ++i) pa.load_part(p.a, m0, mvalid, i);
        } else if (slot == NIA + NIW + 1) {
#pragma unroll
          for (int i = (NIA + 1) / 2; i < NIA; ++i) pa.load_part(p.a, m0, mvalid, i);
        } else if (slot == NIA + NIW + 2) {
#pragma unroll
          for (int i = 0; i < (NIW + 1) / 2; ++i) {
            if (WRED) pwr.load_part(p.W, p.ldw, n0, p.N, k2, p.K, i); else pw.load_part(p.W, p.ldw, n0, p.N, k2, p.K, i);
          }
        } else if (slot == NIA + NIW + 3) {
#pragma unroll
          for (int i = (NIW + 1) / 2; i < NIW; ++i) {
            if (WRED) pwr.load_part(p.W, p.ldw, n0, p.N, k2, p.K, i); else pw.load_part(p.W, p.ldw, n0, p.N, k2, p.K, i);
          }
        }
      };
      if (WRED) spg_mfma_chunk_or_il<TI, TJ>(Ac, reinterpret_cast<const float*>(Bc), IT + 1, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
      else spg_mfma_chunk_il<TI, TJ>(Ac, Bc, IT + 1, JT + 1, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
      __syncthreads();
    }
  } else if constexpr (AMODE == -2) {
    // channel-major clouds (the first convolution of a segment, <= 32 input channels = one chunk): the A tile through the
    // vector pipe (4 channel loads per lane, STN transform applied while staging), the tiny unaligned weight tile scalar
    SpgRowsPipe<SPG_PRO_CLOUD, IT> pa;
    for (int k0 = 0; k0 < p.K; k0 += SPG_KC) {
      pa.load(p.a, m0, mvalid, k0, p.K);
      if (WRED) spg_stage_weight_red<JT>(p.W, p.ldw, n0, p.N, k0, p.K, Bsr);
      else spg_stage_weight<JT>(p.W, p.ldw, n0, p.N, k0, p.K, Bs);
      pa.store(As);
      __syncthreads();
      if (WRED) spg_mfma_chunk_or<TI, TJ>(As, Bsr, IT + 1, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc);
      else spg_mfma_chunk<TI, TJ>(As, Bs, IT + 1, JT + 1, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc);
      __syncthreads();
    }
  } else {
    for (int k0 = 0; k0 < p.K; k0 += SPG_KC) {
      spg_stage_rows<IT>(p.a, m0, mvalid, k0, p.K, As);
      if (WRED) spg_stage_weight_red<JT>(p.W, p.ldw, n0, p.N, k0, p.K, Bsr);
      else spg_stage_weight<JT>(p.W, p.ldw, n0, p.N, k0, p.K, Bs);
      __syncthreads();
      if (WRED) spg_mfma_chunk_or<TI, TJ>(As, Bsr, IT + 1, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc);
      else spg_mfma_chunk<TI, TJ>(As, Bs, IT + 1, JT + 1, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc);
      __syncthreads();
    }
  }
  float* red = reinterpret_cast<float*>(smem);  // LDS is free again after the last barrier
  SpgStatAcc<TJ> none;                          // (unused: one tile per workgroup, nothing accumulated)
  none.has_sg = false;
  spg_tile_epilogue<IT, JT, WI, WJ, WRED, false, false>(p, acc, red, tile, m0, mvalid, n0, none, false);
}

template <int IT, int JT, int WI, int WJ, bool WRED, int AMODE, bool FULL = false, bool STREAM = false, int PREC = 0>
__global__ __launch_bounds__(SPG_THREADS, 2) void spg_rowgemm_kernel(const SpgGemmParams p_by_value) {
  // The parameters are read where they lie -- in the kernel-argument segment (constant address space: scalar loads at the point
  // of use) -- instead of through the by-value struct, which the compiler preloads into SGPRs and then keeps alive across the
  // whole tile loop: the persistent instantiations spilled 74-123 SGPRs to VGPR lanes / scratch that way;
  // with the reference 10-46 (profiles/r05_kernel_resources.txt)
  const SpgGemmParams& p = *(const SpgGemmParams*)(spg_kernarg_ptr<SpgGemmParams>)__builtin_amdgcn_kernarg_segment_ptr();
  spg_touch_params<(int)sizeof(SpgGemmParams)>((spg_kernarg_ptr<unsigned>)__builtin_amdgcn_kernarg_segment_ptr());
  spg_rowgemm_body<IT, JT, WI, WJ, WRED, AMODE, FULL, STREAM, PREC>(p, (int)blockIdx.x, (int)blockIdx.y);
}


// ---------------------------------------------------------------------------------------------
// few-row GEMM with the REDUCTION split over the four waves (round 4)
// ---------------------------------------------------------------------------------------------
// The FC layers over superpoints / edges (M = 1000 ... 5000 rows) are dependent chains of launches whose length is the length
// of ONE workgroup's reduction: with the 32 x 128 tile above every wave owns 32 output columns and walks ALL K / 32 chunks
// (~0.75 us each: 16 MFMAs + LDS staging + a barrier) -- PointNet's 257 -> 256 layer 9 chunks, the classifier 11, the filter
// network's last data gradient (K = 1024) 32: a 24 us launch for 0.6 GFLOP.  Here a workgroup owns a 32 x 32 output tile and its
// four waves split the CHUNKS (wave w takes chunks w, w + 4, ...): the chain is 4x shorter and there are 4x as many workgroups
// (N / 32 column tiles); no LDS staging and no barrier in the loop -- a lane loads its MFMA operands straight from global memory
// (lane (r, h) owns row r / output column r and the 16 reduction indices 16h .. 16h + 15 of the chunk: four 16-byte loads per
// operand) and applies the operand prologue in registers; the four partial accumulators are summed through LDS in wave order
// (deterministic) and wave 0 runs the ordinary tile epilogue.  For K >= 128 (below that the waves would idle).
// AN EXPERIMENT, off by default (spg_tune key 12 = 1 enables it): see launch_gemm_shape.
template <bool WRED, int AMODE>
__device__ __forceinline__ void spg_fewrow_sk_body(const SpgGemmParams& p, const int bx, const int by) {
  extern __shared__ f32x4 smem[];
  float* red = reinterpret_cast<float*>(smem);        // [3][16][64] partial accumulators of waves 1..3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (p.stat_slots != nullptr && bx == 0 && by == 0 && threadIdx.x == 0)
    spg_slots_count_add(p.stat_slots, WRED ? p.n_mask : p.N, p.stat_rows != 0 ? p.stat_rows : (long)p.M);
  if constexpr (!WRED) {
    if (p.fold.slots != nullptr) spg_bn_fold_fwd(p.fold, bx == 0 && by == 0);
  } else {
    if (p.fold_bwd.slots != nullptr) spg_bn_fold_bwd(p.fold_bwd, false);
  }
  const long m0 = (long)bx * 32;
  const int mvalid = (int)min(32L, (long)p.M - m0);
  const int n0 = by * 32;
  const bool rowok = r < mvalid;
  const long row = m0 + (rowok ? r : 0);
  const int col = n0 + r;
  const bool colok = col < p.N;
  const int colc = colok ? col : 0;
  const int K = p.K, nchunk = (K + SPG_KC - 1) / SPG_KC;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int c = wave; c < nchunk; c += 4) {
    const int k0 = c * SPG_KC + 16 * h;
    f32x4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + 4 * q;
      const bool kin = k < K;                      // (a quad is addressable up to the next multiple of 4 of K: padded leading dimensions)
      const int kc = kin ? k : 0;
      const SpgQuad qc = spg_quad_consts<AMODE>(p.a, kc, K);
      SpgRaw raw;
      spg_load_raw1<AMODE>(p.a, row, kc, qc.nvalid, raw);
      a[q] = spg_finish_raw<AMODE>(qc, raw, rowok && kin);
      if constexpr (!WRED) {                       // weights [N, K]: this lane's output column, four consecutive reduction indices
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.W + (long)colc * p.ldw + kc);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[q][e] = (kin && k + e < K) ? w[e] : 0.f;
      } else {                                     // weights [K, N] untransposed: one element per reduction index
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = k + e < K;
          const float w = p.W[(long)(ok ? k + e : 0) * p.ldw + colc];
          b[q][e] = ok ? w : 0.f;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], b[q][e], acc, 0, 0, 0);
  }
  // ---- the four waves' partial sums, in wave order ----
  if (wave != 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[((wave - 1) * 16 + q) * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += red[(w * 16 + q) * 64 + lane];
  f32x16 accs[1][1];
  accs[0][0] = acc;
  SpgStatAcc<1> none;
  none.has_sg = false;
  spg_tile_epilogue<32, 32, 1, 1, WRED, false, false>(p, accs, red, bx, m0, mvalid, n0, none, false);
}

template <bool WRED, int AMODE>
__global__ __launch_bounds__(SPG_THREADS) void spg_fewrow_sk_kernel(const SpgGemmParams p) {
  spg_fewrow_sk_body<WRED, AMODE>(p, (int)blockIdx.x, (int)blockIdx.y);
}
#define SPG_SK_MIN_K 128
#define SPG_SK_LDS (3 * 16 * 64 * sizeof(float))

// ---- weights for the bf16 MFMA modes: split (hi = bf16(w), lo = bf16(w - hi)) and, for the data gradient, transposed ----
__global__ __launch_bounds__(256) void spg_split_weights_kernel(const SpgSplitBatch b) {
  const SpgSplitJob j = b.jobs[blockIdx.y];
  const long ldk = (j.K + 7) & ~7, ldn = (j.N + 7) & ~7;
  const long ufwd = (long)j.N * (ldk / 8), ubwd = j.bwd ? (long)j.K * (ldn / 8) : 0;
  for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < ufwd + ubwd; u += (long)gridDim.x * blockDim.x) {
    float v[8];
    char* out;
    long part;
    if (u < ufwd) {            // forward orientation: 8 consecutive k of output channel n
      const long n = u / (ldk / 8), k0 = 8 * (u % (ldk / 8));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = k0 + e < j.K ? j.W[n * j.ldw + k0 + e] : 0.f;
      out = (char*)j.fwd + (n * ldk + k0) * 2;
      part = (long)j.N * ldk * 2;
    } else {                   // transposed: 8 consecutive n of input channel k
      const long t = u - ufwd, k = t / (ldn / 8), n0 = 8 * (t % (ldn / 8));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = n0 + e < j.N ? j.W[(n0 + e) * j.ldw + k] : 0.f;
      out = (char*)j.bwd + (k * ldn + n0) * 2;
      part = (long)j.K * ldn * 2;
    }
    u32x2 h0, l0, h1, l1;
    spg_split_bf16(f32x4{v[0], v[1], v[2], v[3]}, h0, l0);
    spg_split_bf16(f32x4{v[4], v[5], v[6], v[7]}, h1, l1);
    *reinterpret_cast<uint4*>(out) = uint4{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<uint4*>(out + part) = uint4{l0[0], l0[1], l1[0], l1[1]};
  }
}

int spg_launch_split_weights(const SpgSplitBatch& b, hipStream_t stream) {
  if (b.njobs == 0) return 0;
  hipLaunchKernelGGL(spg_split_weights_kernel, dim3(64, b.njobs), dim3(256), 0, stream, b);
  SPG_LAUNCH_CHECK();
  return 0;
}

int spg_gemm_precision() { return g_tune[SPG_TUNE_PRECISION]; }

int spg_gemm_ntiles(const SpgGemmParams& p) { return spg_cdiv(p.M, p.rows_per_tile); }

// waves along the rows (WI) of the tile shape launch_gemm_shape picks: number of statistics / pooling partials per tile
int spg_gemm_row_waves(int rows_per_tile, int N) {
  if (rows_per_tile <= 32) return 1;
  if (N <= 32) return 4;
  return 2;
}

template <int IT, int JT, int WI, int WJ, bool WRED, int AMODE>
static int launch_gemm_t(const SpgGemmParams& p, hipStream_t stream, int* stat_parts) {
  // A tile (out-major) + weight tile (out-major [8][JT+1] float4 or red-major [32][JT+4] floats), double-buffered;
  // the epilogue reuses the region for its reductions (<= 4*WI*JT floats)
  size_t lds = (size_t)(SPG_KC / 4) * (IT + 1) * sizeof(f32x4) +
               (WRED ? (size_t)SPG_KC * (JT + 4) * sizeof(float) : (size_t)(SPG_KC / 4) * (JT + 1) * sizeof(f32x4));
  if (AMODE >= 0) lds *= 2;
  size_t epi = (size_t)4 * WI * JT * sizeof(float);
  const size_t epi_vec = ((size_t)4 * SPG_EPI_WAVE_FLOATS(IT / WI, JT / WJ) + (size_t)2 * WI * JT) * sizeof(float);
  if (epi < epi_vec) epi = epi_vec;
  if (lds < epi) lds = epi;
  dim3 grid(spg_gemm_ntiles(p), spg_cdiv(p.N, JT));
  if (stat_parts != nullptr) *stat_parts = (int)grid.x * WI;      // one statistics partial per tile and row-wave, unless ...
  const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  // few-row launches inside an open group (spg_gemm.h) become jobs of its one launch
  const bool grouped = IT == 32 && spg_group_accepts(stream);
  ProfScope prof(stream, flops, 0, !grouped);
  prof.r.M = p.M; prof.r.N = p.N; prof.r.K = p.K;
  if constexpr (AMODE >= 0) {
    // whole reduction chunks, no per-element prologue masks, every offset inside 32 bits: fast pipes (rows / output channels
    // of a partial last tile are clamped, their results masked by the epilogue)
    const bool mode_ok = AMODE == SPG_PRO_IDENT || AMODE == SPG_PRO_BNBWD ||
                         (AMODE == SPG_PRO_AFFINE && p.a.c0 != nullptr && p.a.n_affine >= p.K) ||
                         (AMODE == SPG_PRO_POOLBWD && p.a.P == IT && p.rows_per_tile == IT && p.M % IT == 0);
    const bool full = mode_ok && p.K % SPG_KC == 0 && (!WRED || (p.N & 3) == 0) &&
                      (long)IT * p.a.ld < (1L << 29) && (long)(WRED ? SPG_KC : JT) * p.ldw < (1L << 29);
    if (full) {
      SpgGemmParams q = p;
      q.vec_store = p.Y != nullptr && (p.ldy & 3) == 0 && (((uintptr_t)p.Y) & 15) == 0;
      if (WRED)    // the backward epilogue also reads the producer's output and the per-channel constants as quads
        q.vec_store = q.vec_store && (p.N & 3) == 0 && (p.Yp == nullptr || ((p.ldyp & 3) == 0 && (((uintptr_t)p.Yp) & 15) == 0)) &&
                      ((((uintptr_t)p.ms) | ((uintptr_t)p.mt) | ((uintptr_t)p.mmean) | ((uintptr_t)p.mrstd) | ((uintptr_t)p.stat)) & 15) == 0;
      prof.r.tag = SPG_PROF_TAG(1, IT, JT, WRED ? 1 : 0, AMODE, 1);
      // Persistent chunk stream (see the kernel): as many workgroups as the chip holds at once (2 per CU), each walks
      // the row tiles of ONE column tile.  Needs whole tiles everywhere, an even number of reduction chunks (the LDS
      // buffer / register set roles are then the same at every tile start) and a column-tile count dividing the stride.
      q.ntile = (int)grid.x; q.rstride = 1 << 30; q.remap = 0; q.ncol = (int)grid.y;
      const int slots = 2 * spg_num_cus();      // (3 per CU fit the 128x64 forward kernels -- 168 VGPRs, 51 KB LDS -- and change nothing: measured)
      const int ncol = (int)grid.y;
      // opt-in bf16 / split-bf16 MFMA (spg_tune key 7): the caller supplied pre-split weights for this orientation
      const int prec = (IT == 128 && q.Wb != nullptr) ? g_tune[SPG_TUNE_PRECISION] : 0;
      // (the 128-column backward kernels -- two operand streams, four constant arrays -- have no registers left for the
      // stream state: measured slower with it, in fp32 and in the bf16 modes (conv5 dgrad 69 -> 120 us), so they keep one
      // workgroup per tile)
      if (!g_tune[SPG_TUNE_NO_PERSIST] && IT == 128 && !(WRED && JT == 128) && p.rows_per_tile == IT && p.M % IT == 0 && p.N % JT == 0 &&
          (p.K / SPG_KC) % 2 == 0 && (long)grid.x * ncol > slots && slots % (8 * ncol) == 0) {
        q.remap = 1; q.rstride = slots / ncol;
#ifdef SPG_ATTRIBUTION
        q.dbg = WRED ? 0 : g_tune[SPG_TUNE_DBG];      // attribution switches: forward launches only (tools/ build)
#else
        q.dbg = 0;                                    // compiled out of the production library (make ATTRIBUTION=1 for tools/)
#endif
        // ... the persistent workgroups accumulate over their tiles: one partial per workgroup of a column tile and row-wave
#ifndef SPG_ATTRIBUTION
        const int no_stat_accum = 0;
#else
        const int no_stat_accum = g_tune[SPG_TUNE_NO_STAT_ACCUM];
#endif
        q.stat_accum = !no_stat_accum && (p.stat != nullptr || p.stat_slots != nullptr) && (!WRED || (q.vec_store && p.N <= p.n_mask));
        if (q.stat_accum && stat_parts != nullptr) *stat_parts = (int)(q.rstride < q.ntile ? q.rstride : q.ntile);      // one per workgroup
        grid = dim3((unsigned)slots, 1);
        if constexpr (IT == 128 && !(WRED && JT == 128)) {
          if (prec == 3) hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, true, 3>), grid, dim3(SPG_THREADS), lds, stream, q);
          else if (prec == 1) hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, true, 1>), grid, dim3(SPG_THREADS), lds, stream, q);
          else hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, true>), grid, dim3(SPG_THREADS), lds, stream, q);
          SPG_LAUNCH_CHECK();
          return 0;
        }
      }
      if constexpr (IT == 128 && WRED && JT == 128) {      // the one wide shape that is not persistent in fp32
        if (prec != 0) {
          if (prec == 3) hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, false, 3>), grid, dim3(SPG_THREADS), lds, stream, q);
          else hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, false, 1>), grid, dim3(SPG_THREADS), lds, stream, q);
          SPG_LAUNCH_CHECK();
          return 0;
        }
      }
      if constexpr (IT == 32 && spg_gemm_variant(WRED, AMODE, true) >= 0) {
        if (grouped && spg_group_add(SPG_JOB_GEMM, spg_gemm_variant(WRED, AMODE, true), &q, sizeof(q), grid, lds, flops, stream, 2 + 2 * spg_cdiv(p.K, SPG_KC),
                                     [q, grid, lds, stream]() -> int {
                                       hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, false>), grid, dim3(SPG_THREADS), lds, stream, q);
                                       SPG_LAUNCH_CHECK();
                                       return 0;
                                     })) return 0;
      }
      hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE, true, false>), grid, dim3(SPG_THREADS), lds, stream, q);
      SPG_LAUNCH_CHECK();
      return 0;
    }
  }
  prof.r.tag = SPG_PROF_TAG(1, IT, JT, WRED ? 1 : 0, AMODE, 0);
  // The masked pipelines too (a reduction that is no multiple of 32 -- the first convolution's 14 / 11 input channels --, a ragged
  // last row tile): forward tiles that ARE complete leave through the vector store (dwordx4 row segments staged through LDS)
  // instead of 16 dword stores per accumulator block -- the epilogue decides per tile (spg_tile_epilogue)
  SpgGemmParams pv = p;
  pv.vec_store = !WRED && p.Y != nullptr && (p.ldy & 3) == 0 && (((uintptr_t)p.Y) & 15) == 0 && !spg_tune_get(SPG_TUNE_NO_VEC_GENERIC);
  if constexpr (IT == 32 && spg_gemm_variant(WRED, AMODE, false) >= 0) {
    if (grouped && spg_group_add(SPG_JOB_GEMM, spg_gemm_variant(WRED, AMODE, false), &pv, sizeof(pv), grid, lds, flops, stream, 2 + 2 * spg_cdiv(p.K, SPG_KC),
                                 [pv, grid, lds, stream]() -> int {
                                   hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE>), grid, dim3(SPG_THREADS), lds, stream, pv);
                                   SPG_LAUNCH_CHECK();
                                   return 0;
                                 })) return 0;
  }
  hipLaunchKernelGGL((spg_rowgemm_kernel<IT, JT, WI, WJ, WRED, AMODE>), grid, dim3(SPG_THREADS), lds, stream, pv);
  SPG_LAUNCH_CHECK();
  return 0;
}

// few rows, long reduction: the split-K form (spg_fewrow_sk_body) -- stand-alone or as a job of the open group
template <bool WRED, int AMODE>
static int launch_fewrow_sk(const SpgGemmParams& p, hipStream_t stream, int* stat_parts) {
  const dim3 grid(spg_gemm_ntiles(p), spg_cdiv(p.N, 32));
  if (stat_parts != nullptr) *stat_parts = (int)grid.x;
  const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
  const bool grouped = spg_group_accepts(stream);
  ProfScope prof(stream, flops, SPG_PROF_TAG(1, 32, 32, WRED ? 1 : 0, AMODE, 0), !grouped);
  prof.r.M = p.M; prof.r.N = p.N; prof.r.K = p.K;
  auto direct = [p, grid, stream]() -> int {
    hipLaunchKernelGGL((spg_fewrow_sk_kernel<WRED, AMODE>), grid, dim3(SPG_THREADS), SPG_SK_LDS, stream, p);
    SPG_LAUNCH_CHECK();
    return 0;
  };
  if (grouped && spg_group_add(SPG_JOB_GEMM, spg_gemm_sk_variant(WRED, AMODE), &p, sizeof(p), grid, SPG_SK_LDS, flops, stream,
                               2 + 2 * spg_cdiv(spg_cdiv(p.K, SPG_KC), 4), direct))
    return 0;
  return direct();
}

template <bool WRED, int AMODE>
static int launch_gemm_shape(const SpgGemmParams& p, hipStream_t stream, int* sp) {
  if constexpr (AMODE == SPG_PRO_IDENT || (!WRED && AMODE == SPG_PRO_AFFINE) || (WRED && AMODE == SPG_PRO_BNBWD)) {
    // OPT-IN (spg_tune key 12): measured no faster than the 32 x 128 kernel on the step (1.404 vs 1.401 ms, same box,
    // profiles/r04_splitk_experiment.txt) -- a link of an FC chain is dispatch + statistics fold + first loads + epilogue; the
    // chunk loop this form shortens is the smaller part, and without operand prefetch its chunks pay the memory latency each
    // (round 5: also tried as the default for the one long reduction of the step, the data gradient of the filter network's last
    // layer, K = 1024 -- 1.1949 vs 1.1947 ms: the group it rides in is not bounded by it)
    if (spg_tune_get(SPG_TUNE_SPLITK) && p.rows_per_tile == SPG_FC_ROWS && p.K >= SPG_SK_MIN_K && p.pool_out == nullptr)
      return launch_fewrow_sk<WRED, AMODE>(p, stream, sp);
  }
  if (p.rows_per_tile <= 32) return launch_gemm_t<32, 128, 1, 4, WRED, AMODE>(p, stream, sp);   // few rows (FC layers, filter net)
  if (p.N <= 32) return launch_gemm_t<128, 32, 4, 1, WRED, AMODE>(p, stream, sp);
  if (p.N <= 64) return launch_gemm_t<128, 64, 2, 2, WRED, AMODE>(p, stream, sp);
  return launch_gemm_t<128, 128, 2, 2, WRED, AMODE>(p, stream, sp);                                // wider outputs: grid.y column tiles
}

static int spg_launch_gemm_impl(const SpgGemmParams& p, hipStream_t stream, int* stat_parts);
static long group_njobs();
int spg_launch_gemm(const SpgGemmParams& p, hipStream_t stream, int* stat_parts) {
  if (p.stat_slots == nullptr || !spg_slot_sync_active()) return spg_launch_gemm_impl(p, stream, stat_parts);
  // slot-synchronised BatchNorm: the slots this launch adds to are all-reduced behind it -- behind the group's launch when the
  // launch became a job of the open group (the job table grew)
  const long before = group_njobs();
  SPG_TRY(spg_launch_gemm_impl(p, stream, stat_parts));
  const int C = p.w_red ? p.n_mask : p.N;
  return spg_slot_sync_after(p.stat_slots, spg_fold_slot_words(C), stream, group_njobs() > before);
}
static int spg_launch_gemm_impl(const SpgGemmParams& p, hipStream_t stream, int* stat_parts) {
  SPG_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty GEMM");
  SPG_CHECK_ARG(p.rows_per_tile >= 1 && p.rows_per_tile <= 128, "rows_per_tile must be in [1,128]");
  SPG_CHECK_ARG(p.epi == SPG_EPI_FWD || p.Y != nullptr, "backward epilogue needs an output");
  SPG_CHECK_ARG((p.epi == SPG_EPI_BWD) == (p.w_red != 0), "forward epilogue <-> [N,K] weights, backward epilogue <-> [K,N] weights");
  SPG_CHECK_ARG(p.epi != SPG_EPI_FWD || p.stat == nullptr || p.stat_cnt != nullptr, "forward statistics need stat_cnt");
  SPG_CHECK_ARG(p.stat_slots == nullptr || p.stat == nullptr, "statistics slots replace the partials");
  SPG_CHECK_ARG(p.fold.slots == nullptr || (p.epi == SPG_EPI_FWD && !p.w_red), "a statistics fold belongs to a forward launch");
  // vector path: 16-byte aligned rows, and every row addressable up to the next multiple of 4 of its logical width
  // (padded leading dimensions; partial quads are masked through the A operand / the store mask)
  const bool walign = (p.ldw & 3) == 0 && (((uintptr_t)p.W) & 15) == 0 && p.ldw >= (((p.w_red ? p.N : p.K) + 3) & ~3);
  const bool vec = spg_operand_vec_ok(p.a) && walign && p.a.ld >= ((p.K + 3) & ~3);
  const int mode = vec ? p.a.mode : -1;
  if (!p.w_red) {
    if (p.a.mode == SPG_PRO_CLOUD && p.rows_per_tile > 32 && p.N > 32 && p.N <= 64)      // first convolution of a segment
      return launch_gemm_t<128, 64, 2, 2, false, -2>(p, stream, stat_parts);
    switch (mode) {
      case SPG_PRO_IDENT: return launch_gemm_shape<false, SPG_PRO_IDENT>(p, stream, stat_parts);
      case SPG_PRO_AFFINE: return launch_gemm_shape<false, SPG_PRO_AFFINE>(p, stream, stat_parts);
      default: return launch_gemm_shape<false, -1>(p, stream, stat_parts);
    }
  }
  switch (mode) {
    case SPG_PRO_IDENT: return launch_gemm_shape<true, SPG_PRO_IDENT>(p, stream, stat_parts);
    case SPG_PRO_BNBWD: return launch_gemm_shape<true, SPG_PRO_BNBWD>(p, stream, stat_parts);
    case SPG_PRO_POOLBWD: return launch_gemm_shape<true, SPG_PRO_POOLBWD>(p, stream, stat_parts);
    default: return launch_gemm_shape<true, -1>(p, stream, stat_parts);
  }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel: reduction over the rows (points / superpoints / edges / nodes)
// ---------------------------------------------------------------------------------------------
// AMODE / BMODE >= 0: compile-time operand modes, vector + software-pipelined loop; < 0: generic scalar staging.
// PREC (FULL only): 0 fp32 MFMA; 1 / 3 bf16 / split-bf16 MFMA with both operands converted while staging (spg_common.h)
// COLSUM: also the column sums of the `a` operand (bias gradient of a layer without BatchNorm); a compile-time switch so
// that the main loop of every other instantiation (all wide BatchNorm layers) carries no predicate / accumulators for it
template <int IT, int JT, int WI, int WJ, int AMODE, int BMODE, bool FULL = false, int PREC = 0, bool COLSUM = false>
__device__ __forceinline__ void spg_wgrad_body(const SpgWgradParams& p, const int bx, const int by, const int bz) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32;
  static_assert(WI * WJ == 4 && TI >= 1 && TJ >= 1, "4 waves per workgroup");
  extern __shared__ f32x4 smem[];
  float* As = reinterpret_cast<float*>(smem);          // red-major [32][IT + pad]
  float* Bs = As + SPG_KC * (IT + 4);                   // red-major [32][JT + pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  const int split = bx;
  const int i0 = by * IT, j0 = bz * JT;
  const long ms = (long)split * p.rows_per_split;
  const long me = min((long)p.M, ms + p.rows_per_split);
  // BatchNorm backward of the `a` operand's layer: its sums arrive as fixed-point slots and become the constants of the
  // BNBWD / POOLBWD prologue here (every workgroup; spg_gemm.h) -- readable behind the barrier at its end
  if (p.fold.slots != nullptr) spg_bn_fold_bwd(p.fold, bx == 0 && by == 0 && bz == 0);
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // optional by-product: column sums of the `a` operand (bias gradient).  Thread t < IT adds up channel i0 + t of every A
  // tile right after it became visible in LDS (red-major fp32 tile: 32 conflict-free reads per chunk); only the
  // workgroups of the first column tile do it.  (Not available in the bf16 modes, whose tiles are not fp32.)
  static_assert(!COLSUM || PREC == 0, "column sums ride along with the fp32 tiles only");
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = COLSUM && bz == 0 && tid < IT;
  auto colsum_tile = [&](const float* __restrict__ At, int stride) __attribute__((always_inline)) {
    if constexpr (COLSUM) {
      if (do_colsum) {
#pragma unroll
        for (int k = 0; k < SPG_KC; k += 4) {
          csum[0] += At[(k + 0) * stride + tid]; csum[1] += At[(k + 1) * stride + tid];
          csum[2] += At[(k + 2) * stride + tid]; csum[3] += At[(k + 3) * stride + tid];
        }
      }
    }
  };

  if constexpr (AMODE >= 0 && BMODE >= 0 && FULL && PREC != 0) {
    // bf16 layout: planes of 16-byte slots, slot = 8 consecutive ROWS of one channel (SpgRedFastB); same pipeline
    constexpr int BUF4 = (SPG_KC / 4) * (IT + 1 + JT + 1);
    f32x4* A4 = smem;
    f32x4* B4 = smem + (SPG_KC / 4) * (IT + 1);
    SpgRedFastB<AMODE, IT> pa;
    SpgRedFastB<BMODE, JT> pb;
    constexpr int NIA = SpgRedFastB<AMODE, IT>::NI, NIB = SpgRedFastB<BMODE, JT>::NI;
    SpgQuad qa, qb;
    pa.init(p.a, i0, qa);
    pb.init(p.b, j0, qb);
#pragma unroll
    for (int i = 0; i < NIA; ++i) pa.load_part(p.a, ms, i0, i);
#pragma unroll
    for (int i = 0; i < NIB; ++i) pb.load_part(p.b, ms, j0, i);
    pa.template store_half<PREC>(qa, A4, 0); pa.template store_half<PREC>(qa, A4, 1);
    pb.template store_half<PREC>(qb, B4, 0); pb.template store_half<PREC>(qb, B4, 1);
    const long mlast = me - SPG_KC;                 // first row of the last chunk of this split
    {
      const long m1 = ms + SPG_KC <= mlast ? ms + SPG_KC : mlast;
#pragma unroll
      for (int i = 0; i < NIA; ++i) pa.load_part(p.a, m1, i0, i);
#pragma unroll
      for (int i = 0; i < NIB; ++i) pb.load_part(p.b, m1, j0, i);
    }
    int sa[TI], sb[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) sa[i] = spg_swz(wi * (IT / WI) + r + 32 * i);
#pragma unroll
    for (int j = 0; j < TJ; ++j) sb[j] = spg_swz(wj * (JT / WJ) + r + 32 * j);
    __syncthreads();
    int buf = 0;
    for (long m = ms; m < me; m += SPG_KC) {
      f32x4* An = A4 + (buf ^ 1) * BUF4;
      f32x4* Bn = B4 + (buf ^ 1) * BUF4;
      const long m2 = m + 2 * SPG_KC <= mlast ? m + 2 * SPG_KC : mlast;      // clamped: a harmless re-load past the end
      auto piece = [&](int slot) __attribute__((always_inline)) {
        if (slot < 2) {
          pa.template store_half<PREC>(qa, An, slot);
        } else if (slot < 4) {
          pb.template store_half<PREC>(qb, Bn, slot - 2);
        } else if (slot == 4) {
#pragma unroll
          for (int i = 0; i < (NIA + 1) / 2; ++i) pa.load_part(p.a, m2, i0, i);
        } else if (slot == 5) {
#pragma unroll
          for (int i = (NIA + 1) / 2; i < NIA; ++i) pa.load_part(p.a, m2, i0, i);
        } else if (slot == 6) {
#pragma unroll
          for (int i = 0; i < NIB; ++i) pb.load_part(p.b, m2, j0, i);
        }
      };
      spg_mfma_chunk_bf16_il<TI, TJ, PREC>(A4 + buf * BUF4, B4 + buf * BUF4, IT + 1, JT + 1, sa, sb, h, acc, piece);
      __syncthreads();
      buf ^= 1;
    }
  } else if constexpr (AMODE >= 0 && BMODE >= 0 && FULL) {
    // full tiles: the fast pipes (loop-invariant offsets, no masks), same pipeline as below
    constexpr int BUF = SPG_KC * (IT + 4 + JT + 4);
    SpgRedFast<AMODE, IT> pa;
    SpgRedFast<BMODE, JT> pb;
    constexpr int NIA = SpgRedFast<AMODE, IT>::NI, NIB = SpgRedFast<BMODE, JT>::NI;
    static_assert(NIA + NIB + 4 <= SPG_KC / 2, "staging pieces must fit the MFMA slots of a chunk");
    SpgQuad qa, qb;
    pa.init(p.a, i0, qa);
    pb.init(p.b, j0, qb);
#pragma unroll
    for (int i = 0; i < NIA; ++i) pa.load_part(p.a, ms, i0, i);
#pragma unroll
    for (int i = 0; i < NIB; ++i) pb.load_part(p.b, ms, j0, i);
#pragma unroll
    for (int i = 0; i < NIA; ++i) pa.store_part(qa, As, i);
#pragma unroll
    for (int i = 0; i < NIB; ++i) pb.store_part(qb, Bs, i);
    const long mlast = me - SPG_KC;                 // first row of the last chunk of this split
    {
      const long m1 = ms + SPG_KC <= mlast ? ms + SPG_KC : mlast;
#pragma unroll
      for (int i = 0; i < NIA; ++i) pa.load_part(p.a, m1, i0, i);
#pragma unroll
      for (int i = 0; i < NIB; ++i) pb.load_part(p.b, m1, j0, i);
    }
    __syncthreads();
    int buf = 0;
    for (long m = ms; m < me; m += SPG_KC) {
      float* An = As + (buf ^ 1) * BUF;
      float* Bn = Bs + (buf ^ 1) * BUF;
      const long m2 = m + 2 * SPG_KC <= mlast ? m + 2 * SPG_KC : mlast;      // clamped: a harmless re-load past the end
      auto piece = [&](int slot) __attribute__((always_inline)) {
        if (slot < NIA) {
          pa.store_part(qa, An, slot);
        } else if (slot < NIA + NIB) {
          pb.store_part(qb, Bn, slot - NIA);
        } else if (slot == NIA + NIB) {
#pragma unroll
          for (int i = 0; i < (NIA + 1) / 2; ++i) pa.load_part(p.a, m2, i0, i);
        } else if (slot == NIA + NIB + 1) {
#pragma unroll
          for (int i = (NIA + 1) / 2; i < NIA; ++i) pa.load_part(p.a, m2, i0, i);
        } else if (slot == NIA + NIB + 2) {
#pragma unroll
          for (int i = 0; i < NIB; ++i) pb.load_part(p.b, m2, j0, i);
        }
      };
      colsum_tile(As + buf * BUF, IT + 4);
      spg_mfma_chunk_rr_il<TI, TJ>(As + buf * BUF, Bs + buf * BUF, IT + 4, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
      __syncthreads();
      buf ^= 1;
    }
  } else if constexpr (AMODE >= 0 && BMODE >= 0) {
    // software-pipelined like spg_rowgemm_kernel: iteration c = MFMAs of chunk c, in their shadow the finish + LDS write
    // of chunk c+1 and then the global loads of chunk c+2; two LDS buffers, one barrier; branch-free (rows past the end
    // of the split are read from its first row and written as zeros).
    // A thread keeps the same channel quad for every row: the per-channel constants are loaded once.
    constexpr int BUF = SPG_KC * (IT + 4 + JT + 4);
    SpgRedPipe<AMODE, IT> pa;
    SpgRedPipe<BMODE, JT> pb;
    constexpr int NIA = SpgRedPipe<AMODE, IT>::NI, NIB = SpgRedPipe<BMODE, JT>::NI;
    static_assert(NIA + NIB + 4 <= SPG_KC / 2, "staging pieces must fit the MFMA slots of a chunk");
    const SpgQuad qa = spg_quad_consts<AMODE>(p.a, i0 + 4 * pa.quad_of(tid), p.N);
    const SpgQuad qb = spg_quad_consts<BMODE>(p.b, j0 + 4 * pb.quad_of(tid), p.K);
    pa.load(p.a, qa, ms, me, i0);
    pb.load(p.b, qb, ms, me, j0);
    pa.store(qa, As);
    pb.store(qb, Bs);
#pragma unroll
    for (int i = 0; i < NIA; ++i) pa.load_part(p.a, qa, ms + SPG_KC, me, ms, i0, i);
#pragma unroll
    for (int i = 0; i < NIB; ++i) pb.load_part(p.b, qb, ms + SPG_KC, me, ms, j0, i);
    __syncthreads();
    int buf = 0;
    for (long m = ms; m < me; m += SPG_KC) {
      float* An = As + (buf ^ 1) * BUF;
      float* Bn = Bs + (buf ^ 1) * BUF;
      const long m2 = m + 2 * SPG_KC;
      auto piece = [&](int slot) __attribute__((always_inline)) {
        if (slot < NIA) {
          pa.store_part(qa, An, slot);
        } else if (slot < NIA + NIB) {
          pb.store_part(qb, Bn, slot - NIA);
        } else if (slot == NIA + NIB) {
#pragma unroll
          for (int i = 0; i < (NIA + 1) / 2; ++i) pa.load_part(p.a, qa, m2, me, ms, i0, i);
        } else if (slot == NIA + NIB + 1) {
#pragma unroll
          for (int i = (NIA + 1) / 2; i < NIA; ++i) pa.load_part(p.a, qa, m2, me, ms, i0, i);
        } else if (slot == NIA + NIB + 2) {
#pragma unroll
          for (int i = 0; i < (NIB + 1) / 2; ++i) pb.load_part(p.b, qb, m2, me, ms, j0, i);
        } else if (slot == NIA + NIB + 3) {
#pragma unroll
          for (int i = (NIB + 1) / 2; i < NIB; ++i) pb.load_part(p.b, qb, m2, me, ms, j0, i);
        }
      };
      colsum_tile(As + buf * BUF, IT + 4);
      spg_mfma_chunk_rr_il<TI, TJ>(As + buf * BUF, Bs + buf * BUF, IT + 4, JT + 4, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc, piece);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    const int sa = spg_red_stride<IT>(p.a), sb = spg_red_stride<JT>(p.b);
    for (long m = ms; m < me; m += SPG_KC) {
      spg_stage_red<IT>(p.a, m, me, i0, p.N, As);
      spg_stage_red<JT>(p.b, m, me, j0, p.K, Bs);
      __syncthreads();
      colsum_tile(As, sa);
      spg_mfma_chunk_rr<TI, TJ>(As, Bs, sa, sb, wi * (IT / WI) + r, wj * (JT / WJ) + r, h, acc);
      __syncthreads();
    }
  }
  if constexpr (COLSUM) {
    if (do_colsum && i0 + tid < p.N) p.colsum[(long)split * p.N + i0 + tid] = (csum[0] + csum[1]) + (csum[2] + csum[3]);
  }
  // partial tile of this split: wave-uniform base + 32-bit lane offsets; unconditional stores when the tile is full
  float* pb = p.partial + ((long)split * p.N + i0) * p.K + j0;
  const unsigned ldk = (unsigned)p.K;
  const bool full = i0 + IT <= p.N && j0 + JT <= p.K;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int kl = wj * (JT / WJ) + 32 * j + r;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int nl = wi * (IT / WI) + 32 * i + spg_acc_row(q, h);
        const unsigned off = (unsigned)nl * ldk + (unsigned)kl;
        if (full) pb[off] = acc[i][j][q];
        else if (i0 + nl < p.N && j0 + kl < p.K) pb[off] = acc[i][j][q];
      }
  }
}

template <int IT, int JT, int WI, int WJ, int AMODE, int BMODE, bool FULL = false, int PREC = 0, bool COLSUM = false>
__global__ __launch_bounds__(SPG_THREADS) void spg_wgrad_kernel(const SpgWgradParams p) {
  spg_wgrad_body<IT, JT, WI, WJ, AMODE, BMODE, FULL, PREC, COLSUM>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// out[i] = sum_k partial[k][i]: 64 elements x 16 split-groups per workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(1024) void spg_reduce_partials_kernel(const float* __restrict__ partial, int nsplit, long n,
                                                                   float* __restrict__ out) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + tx;
  float s = 0.f;
  if (i < n)
    for (int k = ty; k < nsplit; k += 16) s += partial[(long)k * n + i];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tx];
    out[i] = t;
  }
}

static void wgrad_plan(long M, int N, int K, int* IT, int* JT, int* nsplit, int* rps) {
  int jt = K <= 32 ? 32 : (K <= 64 ? 64 : 128);
  int it;
  if (jt == 32) it = 128;
  else if (jt == 64) it = N <= 64 ? 64 : 128;
  else it = 128;
  const int tiles = spg_cdiv(N, it) * spg_cdiv(K, jt);
  int target = 512 / tiles;
  if (target < 1) target = 1;
  long rows = (M + target - 1) / target;
  rows = ((rows + SPG_KC - 1) / SPG_KC) * SPG_KC;
  if (rows < SPG_KC) rows = SPG_KC;
  *IT = it; *JT = jt; *rps = (int)rows; *nsplit = spg_cdiv(M, rows);
}

constexpr int spg_bwdpair_rows(int ci) { return 4096 / ci; }      // rows per tile of the fused backward pair (spg_bwdpair_kernel)
size_t spg_wgrad_workspace_floats(long M, int N, int K) {
  int it, jt, ns, rps;
  wgrad_plan(M, N, K, &it, &jt, &ns, &rps);
  size_t w = (size_t)ns * N * K;
  const size_t c = (size_t)64 * N;   // also large enough for spg_launch_colsum over N columns
  // (the fused backward pair writes one partial per workgroup, one workgroup per CU: for the two-pass 128 -> 256 form that is a
  //  little more than the split plan's own count -- spg_queue_bwdpair)
  if (N == 256 && K == 128 && M >= spg_bwdpair_rows(K) && M % spg_bwdpair_rows(K) == 0) {
    const long ntile = M / spg_bwdpair_rows(K);
    const size_t w2 = (size_t)(ntile < spg_num_cus() ? ntile : spg_num_cus()) * N * K;
    w = w2 > w ? w2 : w;
  }
  return w > c ? w : c;
}

// Slices and probes (spg_gemm.h: spg_queue_wgrad_leaf): while g_wgrad_slice_n > 0 the launch functions below issue ONLY the
// splits [g_wgrad_slice0, g_wgrad_slice0 + g_wgrad_slice_n) of the weight gradient, as a job of the open group (never as a
// kernel of its own: the stand-alone kernels have no split offset); while g_wgrad_probe is set nothing is launched and the id
// of the grouped body the launch WOULD use is left in g_wgrad_probe_variant (-1: none -- the launch cannot be sliced).
namespace { thread_local int g_wgrad_slice0 = 0, g_wgrad_slice_n = 0, g_wgrad_probe_variant = -1; thread_local bool g_wgrad_probe = false; }

template <int IT, int JT, int WI, int WJ, int AMODE, int BMODE>
static int launch_wgrad_t(const SpgWgradParams& p, int nsplit, hipStream_t stream) {
  const size_t lds = (size_t)((AMODE >= 0 && BMODE >= 0) ? 2 : 1) * SPG_KC * (IT + 4 + JT + 4) * sizeof(float);
  const bool sliced = g_wgrad_slice_n > 0;
  dim3 grid(sliced ? g_wgrad_slice_n : nsplit, spg_cdiv(p.N, IT), spg_cdiv(p.K, JT));
  const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K * (sliced ? (double)g_wgrad_slice_n / nsplit : 1.0);
  // reductions over few rows (FC layers, filter net, recurrent cell) inside an open group become jobs of its one launch; the
  // wide convolutions' weight gradients (performance-critical, own occupancy bounds) never do -- except as slices (above)
  const bool grouped = (sliced || p.M <= SPG_GROUP_MAX_WGRAD_ROWS) && spg_group_accepts(stream);
  ProfScope prof(stream, flops, 0, !grouped && !g_wgrad_probe);
  // -> 1: taken (by the group, or recorded by a probe), 0: launch it directly, < 0: error
  auto try_group = [&](int variant) -> int {
    if (g_wgrad_probe) { g_wgrad_probe_variant = variant; return 1; }
    if (grouped && variant >= 0 && spg_group_add(SPG_JOB_WGRAD, variant, &p, sizeof(p), grid, lds, flops, stream, 2 + 2 * (p.rows_per_split / SPG_KC),
                                                 nullptr, sliced ? g_wgrad_slice0 : 0)) return 1;
    if (sliced) { spg_set_error("a weight-gradient slice needs an open group and a grouped body (variant %d)", variant); return -1; }
    return 0;
  };
#define SPG_TRY_GROUP(v) { const int tg_ = try_group(v); if (tg_ != 0) return tg_ < 0 ? 1 : 0; }
  if constexpr (AMODE >= 0 && BMODE >= 0 && AMODE != SPG_PRO_CLOUD && BMODE != SPG_PRO_CLOUD) {
    auto mode_ok = [](int mode, const SpgOperand& d, int nch) {
      if (mode == SPG_PRO_AFFINE) return d.c0 != nullptr && d.n_affine >= nch;
      if (mode == SPG_PRO_POOLBWD) return d.P % SPG_KC == 0;
      return true;
    };
    const bool full = mode_ok(AMODE, p.a, p.N) && mode_ok(BMODE, p.b, p.K) && p.M % SPG_KC == 0 &&
                      p.rows_per_split % SPG_KC == 0 && p.N % IT == 0 && p.K % JT == 0 &&
                      (long)SPG_KC * p.a.ld < (1L << 29) && (long)SPG_KC * p.b.ld < (1L << 29);
    if (full) {
      prof.r.tag = SPG_PROF_TAG(2, IT, JT, AMODE, BMODE, 1);
      // opt-in bf16 / split-bf16 MFMA (spg_tune key 7): only for launches that asked for it (the PointNet convolutions set
      // allow_lowp; the filter network, the RNN cell and the stand-alone dense layer stay fp32 whatever their shape)
      const int prec = ((IT == 128 || JT >= 64) && p.colsum == nullptr && p.allow_lowp) ? g_tune[SPG_TUNE_PRECISION] : 0;
      if constexpr (AMODE == SPG_PRO_IDENT) {
        if (p.colsum != nullptr) {
          SPG_TRY_GROUP(spg_wgrad_variant(IT, JT, AMODE, BMODE, true, true));
          hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE, true, 0, true>), grid, dim3(SPG_THREADS), lds, stream, p);
          SPG_LAUNCH_CHECK();
          return 0;
        }
      }
      if (prec == 0) SPG_TRY_GROUP(spg_wgrad_variant(IT, JT, AMODE, BMODE, true, false))
      else if (g_wgrad_probe) { g_wgrad_probe_variant = -1; return 0; }
      else if (sliced) { spg_set_error("weight-gradient slices are fp32-MFMA launches"); return 1; }
      if (prec == 3) hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE, true, 3>), grid, dim3(SPG_THREADS), lds, stream, p);
      else if (prec == 1) hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE, true, 1>), grid, dim3(SPG_THREADS), lds, stream, p);
      else
      hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE, true>), grid, dim3(SPG_THREADS), lds, stream, p);
      SPG_LAUNCH_CHECK();
      return 0;
    }
  }
  prof.r.tag = SPG_PROF_TAG(2, IT, JT, AMODE, BMODE, 0);
  if constexpr (AMODE == SPG_PRO_IDENT || AMODE < 0) {
    if (p.colsum != nullptr) {
      SPG_TRY_GROUP(spg_wgrad_variant(IT, JT, AMODE, BMODE, false, true));
      hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE, false, 0, true>), grid, dim3(SPG_THREADS), lds, stream, p);
      SPG_LAUNCH_CHECK();
      return 0;
    }
  }
  SPG_CHECK_ARG(p.colsum == nullptr, "column sums ride along only with an identity `a` operand");
  SPG_TRY_GROUP(spg_wgrad_variant(IT, JT, AMODE, BMODE, false, false));
#undef SPG_TRY_GROUP
  hipLaunchKernelGGL((spg_wgrad_kernel<IT, JT, WI, WJ, AMODE, BMODE>), grid, dim3(SPG_THREADS), lds, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}

template <int AMODE, int BMODE>
static int launch_wgrad_shape(const SpgWgradParams& p, int it, int jt, int ns, hipStream_t stream) {
  if (jt == 32) return launch_wgrad_t<128, 32, 4, 1, AMODE, BMODE>(p, ns, stream);
  if (jt == 64 && it == 64) return launch_wgrad_t<64, 64, 2, 2, AMODE, BMODE>(p, ns, stream);
  if (jt == 64) return launch_wgrad_t<128, 64, 2, 2, AMODE, BMODE>(p, ns, stream);
  return launch_wgrad_t<128, 128, 2, 2, AMODE, BMODE>(p, ns, stream);
}

template <int AMODE>
static int launch_wgrad_b(const SpgWgradParams& p, int it, int jt, int ns, hipStream_t stream) {
  switch (p.b.mode) {
    case SPG_PRO_IDENT: return launch_wgrad_shape<AMODE, SPG_PRO_IDENT>(p, it, jt, ns, stream);
    case SPG_PRO_AFFINE: return launch_wgrad_shape<AMODE, SPG_PRO_AFFINE>(p, it, jt, ns, stream);
    case SPG_PRO_CLOUD:   // raw clouds (first conv of a segment): at most 32 input channels, one column tile
      if (jt == 32) return launch_wgrad_t<128, 32, 4, 1, AMODE, SPG_PRO_CLOUD>(p, ns, stream);
      return launch_wgrad_shape<-1, -1>(p, it, jt, ns, stream);
    default: return launch_wgrad_shape<-1, -1>(p, it, jt, ns, stream);
  }
}

// launches the split-partial kernel; `part` = dW itself when the plan has a single split
static int spg_launch_wgrad_partials(SpgWgradParams p, float* part, hipStream_t stream) {
  int it, jt, ns, rps;
  wgrad_plan(p.M, p.N, p.K, &it, &jt, &ns, &rps);
  p.rows_per_split = rps;
  p.partial = part;
  // the vector path needs whole channel quads inside the matrices (N, K multiples of 4 are guaranteed by padded
  // leading dimensions: quads past the last channel are masked, but must be addressable)
  const bool b_ok = p.b.mode == SPG_PRO_CLOUD ? (p.K <= 32) : (spg_operand_vec_ok(p.b) && p.b.ld >= ((p.K + 3) & ~3));
  const bool vec = spg_operand_vec_ok(p.a) && p.a.ld >= ((p.N + 3) & ~3) && b_ok;
  if (!vec) return launch_wgrad_shape<-1, -1>(p, it, jt, ns, stream);
  if (p.a.mode == SPG_PRO_IDENT) return launch_wgrad_b<SPG_PRO_IDENT>(p, it, jt, ns, stream);
  if (p.a.mode == SPG_PRO_BNBWD) return launch_wgrad_b<SPG_PRO_BNBWD>(p, it, jt, ns, stream);
  if (p.a.mode == SPG_PRO_POOLBWD) return launch_wgrad_b<SPG_PRO_POOLBWD>(p, it, jt, ns, stream);
  return launch_wgrad_shape<-1, -1>(p, it, jt, ns, stream);
}

size_t spg_wgrad_colsum_floats(long M, int N, int K) {
  int it, jt, ns, rps;
  wgrad_plan(M, N, K, &it, &jt, &ns, &rps);
  return (size_t)ns * N;
}

int spg_launch_wgrad(SpgWgradParams p, float* dW, float* work, hipStream_t stream) {
  SPG_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty wgrad");
  p.colsum = nullptr;
  int it, jt, ns, rps;
  wgrad_plan(p.M, p.N, p.K, &it, &jt, &ns, &rps);
  SPG_TRY(spg_launch_wgrad_partials(p, ns == 1 ? dW : work, stream));
  if (ns > 1) {
    const long n = (long)p.N * p.K;
    hipLaunchKernelGGL(spg_reduce_partials_kernel, dim3(spg_cdiv(n, 64)), dim3(1024), 0, stream, work, ns, n, dW);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// deferred (batched) reductions
// ---------------------------------------------------------------------------------------------
struct SpgReduceBatch {
  SpgReduceJob jobs[SPG_MAX_REDUCE_JOBS];
  int first_block[SPG_MAX_REDUCE_JOBS + 1];
  int njobs;
};

// 256 output elements x 16 split-groups per workgroup: a wave reads 1 KiB of one partial per instruction (float4 per
// lane, 4 loads in flight); fixed summation order (group-local sequence, then the 16 groups in order): deterministic
#define SPG_REDUCE_ELEMS 256
__global__ __launch_bounds__(1024) void spg_reduce_batch_kernel(const SpgReduceBatch b) {
  __shared__ f32x4 red[16][64];
  int j = 0;
  while (j + 1 < b.njobs && (int)blockIdx.x >= b.first_block[j + 1]) ++j;      // wave-uniform scan (<= 40 entries)
  const SpgReduceJob job = b.jobs[j];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = ((long)(blockIdx.x - b.first_block[j]) * 64 + tx) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (job.n & 3) == 0 && ((((uintptr_t)job.out) | ((uintptr_t)job.partial)) & 15) == 0;   // wave-uniform
  if (vec) {                         // whole, aligned quads
    if (i < job.n) {
      int k = ty;
      for (; k + 48 < job.nsplit; k += 64) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(job.partial + (long)k * job.n + i);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 16) * job.n + i);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 32) * job.n + i);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 48) * job.n + i);
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; k < job.nsplit; k += 16) s += *reinterpret_cast<const f32x4*>(job.partial + (long)k * job.n + i);
    }
  } else {
    for (int e = 0; e < 4; ++e)
      if (i + e < job.n)
        for (int k = ty; k < job.nsplit; k += 16) s[e] += job.partial[(long)k * job.n + i + e];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < job.n) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tx];      // fixed order: deterministic
    if (vec) {
      *reinterpret_cast<f32x4*>(job.out + i) = t;
    } else {
      for (int e = 0; e < 4; ++e)
        if (i + e < job.n) job.out[i + e] = t[e];
    }
  }
}

namespace { thread_local std::vector<SpgReduceJob> g_deferred_reduce; }
void spg_reduce_defer(const SpgReduceJob& job) { g_deferred_reduce.push_back(job); }
void spg_reduce_deferred_clear() { g_deferred_reduce.clear(); }

static int launch_reduce_jobs(const SpgReduceJob* jobs, int n, hipStream_t stream) {
  for (int base = 0; base < n; base += SPG_MAX_REDUCE_JOBS) {
    const int m = n - base < SPG_MAX_REDUCE_JOBS ? n - base : SPG_MAX_REDUCE_JOBS;
    SpgReduceBatch b;
    int blocks = 0;
    for (int j = 0; j < m; ++j) {
      b.jobs[j] = jobs[base + j];
      b.first_block[j] = blocks;
      blocks += spg_cdiv(jobs[base + j].n, SPG_REDUCE_ELEMS);
    }
    b.first_block[m] = blocks;
    b.njobs = m;
    hipLaunchKernelGGL(spg_reduce_batch_kernel, dim3(blocks), dim3(1024), 0, stream, b);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

int spg_flush_reduce(SpgReduceQueue& q, hipStream_t stream) {
  if (q.njobs == 0 && g_deferred_reduce.empty()) return 0;
  std::vector<SpgReduceJob> all(g_deferred_reduce);      // deferred jobs of this thread travel with this launch
  g_deferred_reduce.clear();
  all.insert(all.end(), q.jobs, q.jobs + q.njobs);
  q.njobs = 0;
  return launch_reduce_jobs(all.data(), (int)all.size(), stream);
}

// the queue's jobs become jobs of the group that is open on this thread (their partials are complete in stream order: the
// launches that wrote them left earlier); the queue is emptied.  No open group: nothing happens, the jobs wait for the flush.
// max_bytes: volume of partials taken per call (jobs in queue order, at least one) -- a reduction that rides should end in the
// shadow of its group's other jobs (~3 TB/s measured: ~35 MB per 12 us group); the rest stays queued for the next call / the flush
int spg_reduce_ride(SpgReduceQueue& q, hipStream_t stream, size_t max_bytes) {
  if (!spg_group_accepts(stream)) return 0;
  SpgRidingScope riding;
  int kept = 0;
  size_t taken = 0;
  bool full = false;
  for (int j = 0; j < q.njobs; ++j) {
    const SpgReduceJob job = q.jobs[j];
    const size_t bytes = (size_t)job.nsplit * job.n * sizeof(float);
    if (!full && taken > 0 && taken + bytes > max_bytes) full = true;
    if (full || !spg_group_add(SPG_JOB_REDUCE, 0, &job, sizeof(job), dim3(spg_cdiv(job.n, 64)), 16 * 16 * sizeof(f32x4), 0.0, stream,
                               2 + job.nsplit / 32)) {
      q.jobs[kept++] = job;
      continue;
    }
    taken += bytes;
  }
  q.njobs = kept;
  return 0;
}

int spg_flush_deferred_reduce(hipStream_t stream) {
  SpgReduceQueue none;
  return spg_flush_reduce(none, stream);
}

// (`used` only ever grows during a queue's life: a flush in the middle of a pass -- job table full -- sums and forgets the JOBS
//  queued so far but never hands out their arena space again, so partial buffers captured by pending leaves (spg_queue_wgrad_leaf)
//  stay theirs until the queue dies; a pass that needs more than the arena fails with an argument error instead of wrapping around)
static int queue_take(SpgReduceQueue& q, size_t floats, float** out, hipStream_t stream) {
  if (q.njobs == SPG_MAX_REDUCE_JOBS) SPG_TRY(spg_flush_reduce(q, stream));
  SPG_CHECK_ARG(q.arena != nullptr && q.used + floats <= q.arena_floats, "reduction arena too small");
  *out = q.arena + q.used;
  q.used += (floats + 63) & ~(size_t)63;
  return 0;
}

// room for `nsplit` partials of `n` floats each in the queue's arena + the job that sums them into `out` (the caller's kernel writes
// the partials: in stream order before the queue is flushed)
int spg_queue_partials(SpgReduceQueue& q, int nsplit, int n, float* out, float** partial, hipStream_t stream) {
  SPG_CHECK_ARG(nsplit >= 1 && n >= 1 && out != nullptr && partial != nullptr, "partials");
  if (q.njobs + 1 > SPG_MAX_REDUCE_JOBS) SPG_TRY(spg_flush_reduce(q, stream));
  SPG_TRY(queue_take(q, (size_t)nsplit * n, partial, stream));
  SpgReduceJob& j = q.jobs[q.njobs++];
  j.partial = *partial; j.out = out; j.nsplit = nsplit; j.n = n;
  return 0;
}

int spg_queue_wgrad(SpgReduceQueue& q, SpgWgradParams p, float* dW, hipStream_t stream, float* db) {
  SPG_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty wgrad");
  int it, jt, ns, rps;
  wgrad_plan(p.M, p.N, p.K, &it, &jt, &ns, &rps);
  float* part = dW;
  if (q.njobs + 2 > SPG_MAX_REDUCE_JOBS) SPG_TRY(spg_flush_reduce(q, stream));
  if (ns > 1) SPG_TRY(queue_take(q, (size_t)ns * p.N * p.K, &part, stream));
  p.colsum = nullptr;
  if (db != nullptr) {          // the bias gradient rides along: column sums of the `a` operand per split
    p.colsum = db;
    if (ns > 1) SPG_TRY(queue_take(q, (size_t)ns * p.N, &p.colsum, stream));
  }
  SPG_TRY(spg_launch_wgrad_partials(p, part, stream));
  if (ns > 1) {
    SpgReduceJob& j = q.jobs[q.njobs++];
    j.partial = part; j.out = dW; j.nsplit = ns; j.n = p.N * p.K;
    if (db != nullptr) {
      SpgReduceJob& c = q.jobs[q.njobs++];
      c.partial = p.colsum; c.out = db; c.nsplit = ns; c.n = p.N;
    }
  }
  return 0;
}

// spg_epilogue_bwd_vec with the producer's raw output read from LDS (the tile the loaders staged: [IT][ldl] floats, all JT
// channels) instead of from global memory: no load latency inside the epilogue -- its four waves are the only ones that run it
// pass (uniform): 0 = the whole layer; 1 = the FIRST half of the layer's output channels of a two-launch pair (spg_queue_bwdpair:
// 128 -> 256, whose weight matrix does not fit LDS): the partial data gradient is stored as it is -- no mask, no sums; 2 = the
// SECOND half: the stored partial of pass 1 is added first (one dwordx4 load per output quad, issued before the staging), then
// mask / sums / store as usual
template <int IT, int JT, int WI, int WJ>
__device__ __forceinline__ void spg_epilogue_bwd_vec_lds(const SpgGemmParams& p, f32x16 (&acc)[IT / WI / 32][JT / WJ / 32],
                                                         float* __restrict__ red, const float* __restrict__ ylds, int ldl, long m0,
                                                         SpgStatAcc<JT / WJ / 32>& sacc, const int pass) {
  constexpr int TI = IT / WI / 32, TJ = JT / WJ / 32, RW = IT / WI, CW = JT / WJ;
  constexpr int LD = CW + 8, LPR = CW / 4, RPI = 64 / LPR, NIT = SPG_EPI_PIECE_ROWS / RPI;
  const int tid = threadIdx.x + spg_opaque_zero(), lane = tid & 63, wave = (tid >> 6) & 3;
  const int r = lane & 31, h = lane >> 5;
  const int wi = wave / WJ, wj = wave % WJ;
  const int colw = wj * CW, roww = wi * RW;
  float* st = red + wave * SPG_EPI_WAVE_FLOATS(RW, CW);
  const int lr = lane / LPR, lc = 4 * (lane % LPR);
  const int col = colw + lc;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(p.ms + col), sh = *reinterpret_cast<const f32x4*>(p.mt + col);
  const f32x4 mean = *reinterpret_cast<const f32x4*>(p.mmean + col), rstd = *reinterpret_cast<const f32x4*>(p.mrstd + col);
  const float* yl = ylds + (roww + lr) * ldl + col;
  float* yo = p.Y + (m0 + roww) * p.ldy + colw;
  const unsigned oo0 = (unsigned)lr * (unsigned)p.ldy + (unsigned)lc;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  f32x4 prev[TI * 2 * NIT];
  if (pass == 2) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int u = 0; u < NIT; ++u)
          prev[(2 * i + half) * NIT + u] = *reinterpret_cast<const f32x4*>(yo + oo0 + (unsigned)(32 * i + 16 * half + RPI * u) * (unsigned)p.ldy);
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int rb = 32 * i + 16 * half;       // first row of this piece inside the wave's sub-tile
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) st[((u & 3) + 8 * (u >> 2) + 4 * h) * LD + 32 * j + r] = acc[i][j][8 * half + u];
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        f32x4 v = *reinterpret_cast<const f32x4*>(st + (lr + RPI * u) * LD + lc);
        if (pass == 2) v += prev[(2 * i + half) * NIT + u];
        if (pass != 1) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(yl + (rb + RPI * u) * ldl);
          const f32x4 act = spg_fma4(yv, sc, sh), xhat = (yv - mean) * rstd;      // (packed: two channels per VALU operation)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!(act[e] > 0.f)) v[e] = 0.f;
          s1 += v;
          s2 = spg_fma4(v, xhat, s2);
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(yo + oo0 + (unsigned)(rb + RPI * u) * (unsigned)p.ldy));      // (streamed: see spg_store_tile_vec_impl)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  sacc.s1 += s1; sacc.s2 += s2;
}

// ---------------------------------------------------------------------------------------------
// backward of a convolution in ONE pass over dz (round 4)
// ---------------------------------------------------------------------------------------------
// The data gradient and the weight gradient of a layer both start from dz = BatchNorm-backward(g, y): as two launches each
// reads g and y from HBM and re-runs the prologue.  For the layers whose whole weight matrix fits LDS next to two tile buffers
// (64 -> 64, 64 -> 128, 128 -> 128) one workgroup of SIXTEEN waves with fixed roles does both from one LDS copy of each tile
// of IT = 4096 / CI rows:
//   waves 8-15 (loaders)  global -> registers -> LDS, three tiles ahead of the matrix waves: while tile t is multiplied they
//                         finish tile t+1 into the OTHER LDS buffer (dz with the prologue applied, out-major; the producer's RAW
//                         output y_prev, red-major) and then issue the loads of tile t+3 (two tiles are always in flight) --
//                         one workgroup barrier per tile
//   waves 0-3             data gradient  g_prev[IT, CI] = dz[IT, CO] W[CO, CI]  (W red-major, resident in LDS for the whole
//                         launch; one 32 x 32 block per wave) and the backward epilogue fed from LDS: ReLU mask of the producer,
//                         vector store, the producer's BatchNorm-backward sums (accumulated over the workgroup's tiles: ONE
//                         slot contribution per workgroup -- no finalize launch behind a 128-column data gradient any more)
//   waves 4-7             weight gradient dW[CO, CI] += dz^T ReLU(s y_prev + t): the SAME dz tile read transposed
//                         (spg_mfma_chunk_tr_aff; scale / shift / ReLU applied to the operand as it is read), accumulated in
//                         registers over all tiles of the workgroup: one partial per workgroup for the batched reduction
// One workgroup per CU (97 ... 149 KB of LDS), tiles b, b + grid, ...; both matrix roles issue CO / 2 MFMAs per wave and tile.
// HBM per tile: g + y + y_prev + g_prev, each ONCE (before: g and y twice, y_prev three times).  Measured on the unit scene
// (profiles/r04_kernel_stats.txt, profiles/r04_ab_bwdpair_step.txt): 64 -> 64  50.7 -> 35.5 us (131 MB: 3.7 TB/s including ~8 us of launch, prologue and tail),
// 64 -> 128  75 -> 57.5 us.  Arithmetic per element is that of the separate kernels (same prologue expressions, fp32 MFMA); the
// summation ORDER of dW and of the statistics differs from theirs (spg_tune key 14 = 1 restores the separate launches).
// The three roles run their own loops with the same number of workgroup barriers (s_barrier counts waves, not code addresses).
struct SpgBwdPairParams {
  SpgGemmParams g;      // the data-gradient problem: a = dz operand (BNBWD / POOLBWD), W [CO, CI], Y, Yp, ms / mt / mmean / mrstd,
                        // stat_slots (producer), fold_bwd (this layer's sums -> constants, dgamma / dbeta; or none), ntile = M / IT
  SpgOperand b;         // the layer's input: AFFINE + ReLU over the producer's raw output (CI channels)
  float* partial;       // [grid][CO][CI]
  int pass;             // 0: the whole layer in this launch; 1 / 2: first / second half of the layer's OUTPUT channels (a layer whose
                        // weight matrix does not fit LDS, 128 -> 256): every pointer of `g` that is indexed by an output channel is
                        // offset by the caller; pass 1 stores the partial data gradient raw, pass 2 adds it, masks, sums, stores;
                        // the BatchNorm-backward fold (all CO channels of the layer) is pass 1's, the statistics contribution pass 2's
  int pad_;
};
#define SPG_PAIR_THREADS 1024
#ifdef SPG_ATTRIBUTION
// (attribution builds only: make ATTRIBUTION=1, tools/bwdpair_timing.py) shader cycles per role and phase, summed over one wave per
// 256 threads of every role of every workgroup: [shape][role][phase]
__device__ unsigned long long spg_pair_role_t[3][3][4];
extern "C" int spg_pair_role_times(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spg_pair_role_t), sizeof(unsigned long long) * 36) != hipSuccess) return -1;
  if (clear) { unsigned long long z[36] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(spg_pair_role_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define SPG_T0() unsigned long long t__ = __builtin_readcyclecounter(), tp__[4] = {0, 0, 0, t__ - tent__}
#define SPG_TENTRY() const unsigned long long tent__ = __builtin_readcyclecounter(); unsigned long long te__ = tent__, tq__[4] = {0, 0, 0, 0}
#define SPG_TQ(k) { const unsigned long long n__ = __builtin_readcyclecounter(); tq__[k] += n__ - te__; te__ = n__; }
#define SPG_TQEND() if (threadIdx.x == 0) { const int sh__ = (CO == 128) + (CI == 128); for (int k__ = 0; k__ < 4; ++k__) atomicAdd(&spg_pair_entry_t[sh__][k__], tq__[k__]); }
__device__ unsigned long long spg_pair_entry_t[3][4];
extern "C" int spg_pair_entry_times(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(spg_pair_entry_t), sizeof(unsigned long long) * 12) != hipSuccess) return -1;
  if (clear) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(spg_pair_entry_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define SPG_TP(k) { const unsigned long long n__ = __builtin_readcyclecounter(); tp__[k] += n__ - t__; t__ = n__; }
#define SPG_TEND(role) if ((threadIdx.x & 255) == 0) { const int sh__ = (CO == 128) + (CI == 128); for (int k__ = 0; k__ < 4; ++k__) atomicAdd(&spg_pair_role_t[sh__][role][k__], tp__[k__]); }
#else
#define SPG_T0()
#define SPG_TENTRY()
#define SPG_TQ(k)
#define SPG_TQEND()
#define SPG_TP(k)
#define SPG_TEND(role)
#endif
template <int CO, int CI>
constexpr size_t spg_bwdpair_lds_bytes() {
  return 2 * ((size_t)(CO / 4) * (spg_bwdpair_rows(CI) + 1) * 16 + (size_t)spg_bwdpair_rows(CI) * (CI + 4) * 4) + (size_t)CO * (CI + 4) * 4 +
         (size_t)4 * SPG_EPI_WAVE_FLOATS(32, 32) * 4 + (size_t)(CO + CI / 2) * 16;
}

template <int CO, int CI, int AMODE>
__global__ __launch_bounds__(SPG_PAIR_THREADS) void spg_bwdpair_kernel(const SpgBwdPairParams p) {
  constexpr int IT = spg_bwdpair_rows(CI), SA = IT + 1, SX = CI + 4;
  constexpr int CQ = CO / 4, XQ = CI / 4;
  constexpr int NDZ = IT * CQ / 512, NX = IT * XQ / 512;          // quads per loader thread
  constexpr int RDZ = 512 / CQ, RX = 512 / XQ;                    // rows one pass of the 512 loader threads covers
  constexpr int WJ = CI / 32 > 4 ? 4 : CI / 32, WI = 4 / WJ;      // wave grid of both matrix roles (columns = input channels)
  constexpr int TIW = CO / 32 / WI;                               // 32 x 32 blocks of dW per weight-gradient wave
  constexpr int BUF4 = CQ * SA + IT * SX / 4;                     // float4 slots of one LDS tile buffer (dz, then y_prev)
  constexpr int TPG = 128 / IT;                                   // POOLBWD: tiles per group (P = 128 rows)
  static_assert((CO == 64 || CO == 128) && (CI == 64 || CI == 128) && IT * WJ * 32 == 4096 * (WJ * 32 / CI), "supported shapes");
  static_assert(NDZ >= 1 && NX >= 1 && IT % RDZ == 0 && IT % RX == 0, "loader map");
  SPG_TENTRY();
  extern __shared__ f32x4 smem[];
  float* wl = reinterpret_cast<float*>(smem + 2 * BUF4);          // [CO][CI + 4]: red-major W (whole launch)
  float* red = wl + CO * SX;                                      // epilogue staging of the data-gradient waves
  f32x4* kst = reinterpret_cast<f32x4*>(red + 4 * SPG_EPI_WAVE_FLOATS(32, 32));      // [4][CQ] dz constants, [2][XQ] y_prev constants
  const SpgGemmParams& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the role branches below are uniform
  const int r = lane & 31, h = lane >> 5;
  const int role = wave >> 2;                                     // 0 data gradient, 1 weight gradient, 2 / 3 loaders
  const int wi = (wave & 3) / WJ, wj = (wave & 3) % WJ;
  const int ntile = g.ntile, stride = (int)gridDim.x;
  int tile = (int)blockIdx.x;                     // < ntile (host)

  // what every role does before its loop: (loaders: the first two tiles' loads, below) -- the constants of the dz prologue,
  // finished here from the layer's slots (ends with a barrier) or already there (finalize launch) -- constants and W into LDS
  auto prologue = [&](auto wload) __attribute__((always_inline)) {
    // W does not depend on the fold: its loads are issued in front of it (round 6: they used to start behind the fold's barrier --
    // two global round trips in a row at the head of every pair launch).  By the 512 threads of the matrix roles: their registers
    // are free here, the loaders' hold two tiles in flight.
    constexpr bool WLOAD = decltype(wload)::value;
    constexpr int NWQ = CO * XQ / 512;
    static_assert(CO * XQ % 512 == 0, "whole quads per thread");
    f32x4 wq[NWQ];
    if constexpr (WLOAD) {
#pragma unroll
      for (int k = 0; k < NWQ; ++k) {
        const int i = tid + k * 512, co = i / XQ, q = i % XQ;
        wq[k] = *reinterpret_cast<const f32x4*>(g.W + (long)co * g.ldw + 4 * q);
      }
    }
    if (g.stat_slots != nullptr && p.pass != 1 && blockIdx.x == 0 && tid == 0) spg_slots_count_add(g.stat_slots, g.n_mask, g.stat_rows != 0 ? g.stat_rows : (long)g.M);
    SPG_TQ(0);
    if (g.fold_bwd.slots != nullptr) spg_bn_fold_bwd(g.fold_bwd, blockIdx.x == 0);
    SPG_TQ(1);
    if (tid < CQ) {
      kst[0 * CQ + tid] = *reinterpret_cast<const f32x4*>(g.a.c0 + 4 * tid); kst[1 * CQ + tid] = *reinterpret_cast<const f32x4*>(g.a.c1 + 4 * tid);
      kst[2 * CQ + tid] = *reinterpret_cast<const f32x4*>(g.a.c2 + 4 * tid); kst[3 * CQ + tid] = *reinterpret_cast<const f32x4*>(g.a.c3 + 4 * tid);
    }
    if (tid < XQ) {
      kst[4 * CQ + tid] = *reinterpret_cast<const f32x4*>(p.b.c0 + 4 * tid); kst[4 * CQ + XQ + tid] = *reinterpret_cast<const f32x4*>(p.b.c1 + 4 * tid);
    }
    if constexpr (WLOAD) {
#pragma unroll
      for (int k = 0; k < NWQ; ++k) {
        const int i = tid + k * 512, co = i / XQ, q = i % XQ;
        *reinterpret_cast<f32x4*>(wl + co * SX + 4 * q) = wq[k];
      }
    }
    __syncthreads();                              // constants and W are in LDS
    SPG_TQ(2);
  };

  if (role >= 2) {
    // ---------------- loaders (8 waves) ----------------
    // TWO tiles in flight per workgroup (register sets R0 / R1, tiles k+2 and k+3 while tile k is multiplied)
    const int lt = tid - 512;
    const int cq = lt % CQ, rdz = lt / CQ;        // this thread's channel quad of dz (fixed) and its first row
    const int xq = lt % XQ, rx = lt / XQ;
    // byte offsets of this thread's quads inside a tile (32-bit; the tile's base pointer is wave-uniform)
    const unsigned odz = ((unsigned)rdz * (unsigned)g.a.ld + 4u * (unsigned)cq) * 4u, sdz = (unsigned)RDZ * (unsigned)g.a.ld * 4u;
    const unsigned ox = ((unsigned)rx * (unsigned)p.b.ld + 4u * (unsigned)xq) * 4u, sx = (unsigned)RX * (unsigned)p.b.ld * 4u;
    struct Regs {
      f32x4 pg[AMODE == SPG_PRO_POOLBWD ? 1 : NDZ], py[NDZ], px[NX];
      int4 pai;
    };
    Regs R0, R1;
    auto load_tile = [&](Regs& R, int t) __attribute__((always_inline)) {
      const long m0 = (long)t * IT;
      if constexpr (AMODE == SPG_PRO_POOLBWD) {   // a group (P = 128 rows) is TPG tiles: its pooled gradient / arg-max rows, this thread's quad
        R.pg[0] = spg_ld16(g.a.X + (long)(t / TPG) * g.a.ldg, 16u * (unsigned)cq);
        R.pai = spg_ld16i(g.a.aidx + (long)(t / TPG) * g.a.ldg, 16u * (unsigned)cq);
      } else {
        const float* gb = g.a.X + m0 * g.a.ld;
#pragma unroll
        for (int i = 0; i < NDZ; ++i) R.pg[i] = spg_ld16(gb, odz + sdz * i);
      }
      const float* yb = g.a.X2 + m0 * g.a.ld;
#pragma unroll
      for (int i = 0; i < NDZ; ++i) R.py[i] = spg_ld16(yb, odz + sdz * i);
      const float* xb = p.b.X + m0 * p.b.ld;
#pragma unroll
      for (int i = 0; i < NX; ++i) R.px[i] = spg_ld16(xb, ox + sx * i);
    };
    auto store_tile = [&](const Regs& R, int t, int buf) __attribute__((always_inline)) {
      f32x4* dz4 = smem + buf * BUF4;
      float* xs = reinterpret_cast<float*>(dz4 + CQ * SA);
      const f32x4 ka = kst[0 * CQ + cq], kb = kst[1 * CQ + cq], kc = kst[2 * CQ + cq], kd = kst[3 * CQ + cq];
      const int prow = (t % TPG) * IT;            // POOLBWD: first row of this tile inside its group
#pragma unroll
      for (int i = 0; i < NDZ; ++i) {
        const int row = rdz + RDZ * i;
        f32x4 gv;
        if constexpr (AMODE == SPG_PRO_POOLBWD) {
          gv[0] = R.pai.x == prow + row ? R.pg[0][0] : 0.f; gv[1] = R.pai.y == prow + row ? R.pg[0][1] : 0.f;
          gv[2] = R.pai.z == prow + row ? R.pg[0][2] : 0.f; gv[3] = R.pai.w == prow + row ? R.pg[0][3] : 0.f;
        } else {
          gv = R.pg[i];
        }
        dz4[cq * SA + row] = spg_bnbwd_value4(ka, gv, kb, R.py[i], kc, kd);
      }
      // the producer's RAW output: the weight-gradient waves apply scale / shift / ReLU when they read their operand, the
      // data-gradient waves' epilogue needs the raw value (ReLU mask, xhat)
#pragma unroll
      for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(xs + (rx + RX * i) * SX + 4 * xq) = R.px[i];
    };
    // (no constant needed yet: in flight under the prologue.  The widest shape -- 128 output, 64 input channels: 44 registers per
    //  set -- keeps only ONE set in flight across the prologue: with both, the fold's arithmetic spilled 24 registers of in-flight
    //  loads, i.e. waited for them, and the launch took 25 900 cycles from its first instruction to its first tile instead of 14 000)
    constexpr bool LATE_R1 = 2 * sizeof(Regs) / 4 >= 64;
    load_tile(R0, tile);
    if (!LATE_R1 && tile + stride < ntile) load_tile(R1, tile + stride);
    prologue(std::false_type{});
    if (LATE_R1 && tile + stride < ntile) load_tile(R1, tile + stride);
    store_tile(R0, tile, 0);
    if (tile + 2 * stride < ntile) load_tile(R0, tile + 2 * stride);
    __syncthreads();
    SPG_T0();
    for (;;) {
      // tile k is being multiplied from buffer 0: tile k+1 (R1) -> buffer 1, then the loads of tile k+3 into R1
      int nxt = tile + stride;
      bool has_next = nxt < ntile;                // uniform
      if (has_next) {
        store_tile(R1, nxt, 1);
        SPG_TP(0);
        if (nxt + 2 * stride < ntile) load_tile(R1, nxt + 2 * stride);
        SPG_TP(1);
      }
      __syncthreads();
      SPG_TP(2);
      if (!has_next) break;
      tile = nxt;
      // tile k+1 from buffer 1: tile k+2 (R0) -> buffer 0, then the loads of tile k+4 into R0
      nxt = tile + stride;
      has_next = nxt < ntile;
      if (has_next) {
        store_tile(R0, nxt, 0);
        SPG_TP(0);
        if (nxt + 2 * stride < ntile) load_tile(R0, nxt + 2 * stride);
        SPG_TP(1);
      }
      __syncthreads();
      SPG_TP(2);
      if (!has_next) break;
      tile = nxt;
    }
    __syncthreads();
    SPG_TEND(2);
    return;
  }

  prologue(std::true_type{});
  if (role == 0) {
    // ---------------- data gradient ----------------
    f32x16 acc[1][1];
    SpgStatAcc<1> sacc;
    sacc.n = 0.f; sacc.a[0] = 0.f; sacc.b[0] = 0.f;
    sacc.s1 = f32x4{0.f, 0.f, 0.f, 0.f}; sacc.s2 = sacc.s1;
    __syncthreads();
    SPG_TQ(3); SPG_TQEND();
    int buf = 0;
    SPG_T0();
    for (;;) {
      const int nxt = tile + stride;
      const bool has_next = nxt < ntile;
      const f32x4* dz4 = smem + buf * BUF4;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0][0][q] = 0.f;
      spg_mfma_tile_or<CO / 8, 2>(dz4, wl, SA, SX, wi * 32 + r, wj * 32 + r, h, acc[0][0]);
      SPG_TP(0);
      spg_epilogue_bwd_vec_lds<IT, CI, WI, WJ>(g, acc, red, reinterpret_cast<const float*>(dz4 + CQ * SA), SX, (long)tile * IT, sacc, p.pass);
      SPG_TP(1);
      __syncthreads();
      SPG_TP(2);
      if (!has_next) break;
      tile = nxt; buf ^= 1;
    }
    SPG_TEND(0);
    if (p.pass == 1) return;                      // (uniform) first half: nothing to contribute yet
    // the workgroup's ONE statistics contribution (as at the end of a persistent data-gradient stream)
    constexpr int CW = 32, LPR = CW / 4;
    float* xch = red;                             // [2][CI][2]
    f32x4 s1 = sacc.s1, s2 = sacc.s2;
    const int cl = wj * CW + 4 * (lane % LPR);
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
    }
    if (WI > 1 && wi != 0 && lane < LPR) {
      *reinterpret_cast<f32x4*>(xch + (wi * CI + cl) * 2) = s1;
      *reinterpret_cast<f32x4*>(xch + (wi * CI + cl) * 2 + 4) = s2;
    }
    __syncthreads();
    if (wi == 0 && lane < LPR) {
      if (WI > 1) {
        s1 += *reinterpret_cast<const f32x4*>(xch + (1 * CI + cl) * 2);
        s2 += *reinterpret_cast<const f32x4*>(xch + (1 * CI + cl) * 2 + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) spg_slots_add(g.stat_slots, g.n_mask, cl + e, (double)s1[e], (double)s2[e]);
    }
    return;
  }

  // ---------------- weight gradient ----------------
  f32x16 acc[TIW][1];
#pragma unroll
  for (int i = 0; i < TIW; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][0][q] = 0.f;
  // this lane's input channel is fixed: BatchNorm scale / shift of the producer applied (+ ReLU) to every B operand it reads
  const float xsc = reinterpret_cast<const float*>(kst + 4 * CQ)[wj * 32 + r], xsh = reinterpret_cast<const float*>(kst + 4 * CQ + XQ)[wj * 32 + r];
  __syncthreads();
  int buf = 0;
  SPG_T0();
  for (;;) {
    const int nxt = tile + stride;
    const bool has_next = nxt < ntile;
    const float* dzf = reinterpret_cast<const float*>(smem + buf * BUF4);
    const float* xs = dzf + CQ * SA * 4;
    spg_mfma_chunk_tr_aff<TIW, TIW, IT / 2>(dzf, xs, SA, SX, 0, wi * (32 * TIW) + r, wj * 32 + r, h, xsc, xsh, acc);
    SPG_TP(0);
    __syncthreads();
    SPG_TP(2);
    if (!has_next) break;
    tile = nxt; buf ^= 1;
  }
  SPG_TEND(1);
  __syncthreads();
  float* pb = p.partial + (long)blockIdx.x * CO * CI;
  const int kl = wj * 32 + r;
#pragma unroll
  for (int i = 0; i < TIW; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int nl = wi * (32 * TIW) + 32 * i + spg_acc_row(q, h);
      __builtin_nontemporal_store(acc[i][0][q], pb + nl * CI + kl);      // (streamed: read once, by the step's final reduction)
    }
}

bool spg_bwdpair_supported(const SpgGemmParams& g, const SpgOperand& b) {
  if (g_tune[SPG_TUNE_NO_BWD_PAIR]) return false;      // (also in the opt-in precision modes: the fused pair computes in fp32 MFMA)
  const int CI = g.N, CO = g.K;
  // (CI = 128, CO = 256 -- the pooled layer of the S3DIS / Semantic3D PointNet: two launches over the halves of its output channels)
  if (!(g.w_red && g.epi == SPG_EPI_BWD && ((CI == 64 && (CO == 64 || CO == 128)) || (CI == 128 && (CO == 128 || (CO == 256 && !g_tune[SPG_TUNE_NO_PAIR_SPLIT] && g_tune[SPG_TUNE_PRECISION] == 0)))))) return false;
  // (not in the opt-in bf16 modes: there the pooled layer's stand-alone weight / data gradient run split-bf16 MFMA -- 50 + 65 us at the
  //  unit scene's size against 167 us for the fp32 pair -- measured: Semantic3D scale split-bf16 1.44 -> 1.38 M superpoints/s with it)
  const int IT = spg_bwdpair_rows(CI);
  if (g.M < IT || g.M % IT != 0) return false;
  const SpgOperand& a = g.a;
  if (a.mode != SPG_PRO_BNBWD && a.mode != SPG_PRO_POOLBWD) return false;
  if (!spg_operand_vec_ok(a) || a.ld < CO) return false;
  if (a.mode == SPG_PRO_POOLBWD && (a.P != 128 || a.ldg < CO)) return false;
  if (CO == 256 && g.fold_bwd.slots != nullptr && g.fold_bwd.C != CO) return false;      // (pass 1 finishes the constants of ALL channels)
  if (b.mode != SPG_PRO_AFFINE || !b.relu || b.c0 == nullptr || b.n_affine != CI || !spg_operand_vec_ok(b) || b.ld < CI) return false;
  if ((g.ldw & 3) != 0 || (((uintptr_t)g.W) & 15) != 0 || g.ldw < CI) return false;
  if (g.Y == nullptr || g.Yp == nullptr || (g.ldy & 3) != 0 || (g.ldyp & 3) != 0 || ((((uintptr_t)g.Y) | ((uintptr_t)g.Yp)) & 15) != 0) return false;
  if (g.Yp != b.X || g.ldyp != b.ld) return false;      // the epilogue's producer output IS the layer's input (read once, from LDS)
  if (g.stat_slots == nullptr || g.mmean == nullptr || g.mrstd == nullptr || g.ms == nullptr || g.mt == nullptr) return false;
  if (g.ms != b.c0 || g.mt != b.c1) return false;
  if (!g.mask_relu || g.n_mask != CI) return false;
  if ((((uintptr_t)g.ms) | ((uintptr_t)g.mt) | ((uintptr_t)g.mmean) | ((uintptr_t)g.mrstd)) & 15) return false;
  const long ntile = g.M / IT;
  const long grid = ntile < spg_num_cus() ? ntile : spg_num_cus();
  return (size_t)grid * CO * CI <= spg_wgrad_workspace_floats(g.M, CO, CI);      // the layer's slice of the reduction arena
}

template <int CO, int CI, int AMODE>
static int launch_bwdpair_t(const SpgBwdPairParams& p, int grid, hipStream_t stream) {
  static bool attr_done = false;
  const size_t lds = spg_bwdpair_lds_bytes<CO, CI>();
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spg_bwdpair_kernel<CO, CI, AMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { spg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  hipLaunchKernelGGL((spg_bwdpair_kernel<CO, CI, AMODE>), dim3(grid), dim3(SPG_PAIR_THREADS), lds, stream, p);
  SPG_LAUNCH_CHECK();
  return 0;
}
template <int AMODE>
static int launch_bwdpair_shape(const SpgBwdPairParams& p, int grid, hipStream_t stream) {
  if (p.g.N == 64 && p.g.K == 64) return launch_bwdpair_t<64, 64, AMODE>(p, grid, stream);
  if (p.g.N == 64) return launch_bwdpair_t<128, 64, AMODE>(p, grid, stream);
  return launch_bwdpair_t<128, 128, AMODE>(p, grid, stream);      // (also each half of a 128 -> 256 layer: p.pass 1 / 2)
}

// g: the data-gradient problem as for spg_launch_gemm (w_red = 1, epi = BWD, stat_slots, fold_bwd = this layer's pending sums or none);
// b: the layer's input operand; dW [g.K, g.N].  The caller checks spg_bwdpair_supported first.
int spg_queue_bwdpair(SpgReduceQueue& q, SpgGemmParams g, const SpgOperand& b, float* dW, hipStream_t stream) {
  SPG_CHECK_ARG(spg_bwdpair_supported(g, b), "shape not supported by the fused backward");
  const int IT = spg_bwdpair_rows(g.N);
  const int ntile = g.M / IT;
  const int grid = ntile < spg_num_cus() ? ntile : spg_num_cus();
  g.ntile = ntile; g.vec_store = 1; g.rows_per_tile = IT;
  // 128 -> 256 (the pooled layer): W (256 x 132 floats = 135 KB) does not fit LDS next to the tile buffers -- two launches of the
  // 128 -> 128 kernel over the halves of the OUTPUT channels.  Each reads its half of (g, y) once and the layer's input once
  // (twice in total: 65 MB more than a single pass would), its half of dW is complete; the data gradient is a sum over ALL output
  // channels: pass 1 stores its partial, pass 2 adds it in its epilogue (65 MB more) and then masks / sums / stores.  Against the
  // separate weight- and data-gradient launches (each applies the prologue to all of (g, y)): 0.52 -> 0.33 GB fetched, one launch
  // chain of 2 instead of 3 (no finalize launch: the statistics leave through the slots).
  const int npass = g.K == 256 ? 2 : 1, COh = g.K / npass;
  for (int h = 0; h < npass; ++h) {
    SpgBwdPairParams p;
    memset(&p, 0, sizeof(p));
    p.g = g; p.b = b;
    p.pass = npass == 1 ? 0 : h + 1;
    if (npass == 2) {
      p.g.K = COh;
      p.g.W = g.W + (long)h * COh * g.ldw;
      p.g.a.X += h * COh; p.g.a.X2 += h * COh;                    // dz operand: channels [h COh, (h + 1) COh) of (g | pooled gradient, y)
      if (p.g.a.aidx != nullptr) p.g.a.aidx += h * COh;
      p.g.a.c0 += h * COh; p.g.a.c1 += h * COh; p.g.a.c2 += h * COh; p.g.a.c3 += h * COh;
      if (h == 1) memset(&p.g.fold_bwd, 0, sizeof(p.g.fold_bwd));      // (pass 1 finished the constants of all channels)
    }
    float* dWh = dW + (long)h * COh * g.N;
    if (q.njobs + 1 > SPG_MAX_REDUCE_JOBS) SPG_TRY(spg_flush_reduce(q, stream));
    float* part = dWh;
    if (grid > 1) SPG_TRY(queue_take(q, (size_t)grid * COh * g.N, &part, stream));
    p.partial = part;
    {
      const double flops = 4.0 * (double)g.M * (double)COh * (double)g.N;      // data gradient + weight gradient
      ProfScope prof(stream, flops, SPG_PROF_TAG(4, 64, 64, g.a.mode, SPG_PRO_AFFINE, 1));
      prof.r.M = g.M; prof.r.N = g.N; prof.r.K = COh;
      if (g.a.mode == SPG_PRO_BNBWD) SPG_TRY(launch_bwdpair_shape<SPG_PRO_BNBWD>(p, grid, stream));
      else SPG_TRY(launch_bwdpair_shape<SPG_PRO_POOLBWD>(p, grid, stream));
    }
    if (grid > 1) {
      SpgReduceJob& j = q.jobs[q.njobs++];
      j.partial = part; j.out = dWh; j.nsplit = grid; j.n = COh * g.N;
    }
  }
  SPG_TRY(spg_slot_sync_after(g.stat_slots, spg_fold_slot_words(g.n_mask), stream, false));      // (slot-synchronised BatchNorm)
  return 0;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics
// ---------------------------------------------------------------------------------------------
// ---- BatchNorm forward statistics ------------------------------------------------------------------------------
// Partials (mean, M2) of ntile*wi row groups -> batch mean / rstd / scale / shift.  Single pass in fp64:
//   mean = sum n_b m_b / M ;  M2 = sum M2_b + sum n_b m_b^2 - M mean^2   (Chan et al.; torch CPU BatchNorm also
//   accumulates float statistics in double).
// Geometry: 16 channels x 64 partial-groups per workgroup, grid (N/16, SPG_FIN_SLICES): every slice reduces its share
// of the partials (tree: wave shuffles, then 16 waves via LDS) and publishes an fp64 triple; the LAST slice to arrive
// (agent-scope release / acquire around one relaxed ticket, cdna guide G16) combines the slices in fixed order, so the
// result is deterministic, and resets the ticket for the next launch.
#define SPG_FIN_SLICES 8

static int* g_fin_counters[SPG_MAX_DEVICES] = {nullptr};    // per device: 4096 self-resetting tickets (the only device memory the library owns)
static unsigned g_fin_next[SPG_MAX_DEVICES] = {0};
static std::mutex g_fin_mutex;           // host threads may drive different streams / devices

static int* spg_fin_counter_window(int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SPG_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_fin_mutex);
  if (g_fin_counters[dev] == nullptr) {     // allocated on the device the launch goes to (the current one)
    if (hipMalloc((void**)&g_fin_counters[dev], 4096 * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(g_fin_counters[dev], 0, 4096 * sizeof(int)) != hipSuccess) return nullptr;
  }
  if (g_fin_next[dev] + n > 4096) g_fin_next[dev] = 0;
  int* w = g_fin_counters[dev] + g_fin_next[dev];     // rotating windows: concurrent launches on other streams do not collide
  g_fin_next[dev] += n;
  return w;
}

// (sum n_b m_b, sum n_b m_b^2, sum M2_b, count) of one channel -> the BatchNorm constants + running statistics
// the per-channel parameters a finish needs, fetched by their thread at the START of the kernel: the loads then overlap the
// reduction instead of adding a dependent memory round trip behind it
struct SpgBnChannel { float gamma, beta, rm, rv; };
__device__ __forceinline__ SpgBnChannel spg_bn_channel(int c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* running_mean, const float* running_var) {
  SpgBnChannel ch;
  ch.gamma = gamma ? gamma[c] : 1.f; ch.beta = beta ? beta[c] : 0.f;
  ch.rm = running_mean ? running_mean[c] : 0.f; ch.rv = running_mean ? running_var[c] : 0.f;
  return ch;
}

__device__ __forceinline__ void spg_bn_finish(double a0, double a1, double a2, double M, int c, const SpgBnChannel& ch,
                                              float* running_mean, float* running_var, float momentum, float eps,
                                              int update_times, float* mean_o, float* rstd_o, float* s_o, float* t_o) {
  const double mean = a0 / M;
  double m2 = a2 + a1 - M * mean * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / M;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double g = (double)ch.gamma, be = (double)ch.beta;
  mean_o[c] = (float)mean;
  rstd_o[c] = (float)rstd;
  s_o[c] = (float)(g * rstd);
  t_o[c] = (float)(be - mean * g * rstd);
  if (running_mean != nullptr && update_times > 0) {
    const double uvar = M > 1.0 ? m2 / (M - 1.0) : var;
    float rm = ch.rm, rv = ch.rv;
    for (int u = 0; u < update_times; ++u) {
      rm = (1.f - momentum) * rm + momentum * (float)mean;
      rv = (1.f - momentum) * rv + momentum * (float)uvar;
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

// ---- synchronised BatchNorm (data-parallel ranks normalise over the union of their batches) --------------------
// The library has no communicator of its own: the host registers an all-reduce(sum) over a caller-owned fp64 device
// buffer (spg_set_bn_allreduce, include/spg_hip.h).  Every train-mode BatchNorm then runs as
//   reduce the local partials into the buffer -> all-reduce on the same stream -> finish from the global sums.
struct SpgSyncBn {
  spg_allreduce_fn fn = nullptr;
  void* ctx = nullptr;
  double* buf = nullptr;
  long ndoubles = 0;
};
static SpgSyncBn g_sync;

extern "C" int spg_set_bn_allreduce(spg_allreduce_fn fn, void* ctx, double* buf, long buf_doubles) {
  if (fn != nullptr) {
    SPG_CHECK_ARG(buf != nullptr && buf_doubles >= 4, "synchronised BatchNorm needs a device buffer");
    SPG_CHECK_ARG(!spg_slot_sync_active(), "the finalize-based synchronised BatchNorm and the slot-synchronised mode (spg_set_slot_allreduce) exclude each other");
  }
  g_sync.fn = fn; g_sync.ctx = ctx; g_sync.buf = fn ? buf : nullptr; g_sync.ndoubles = fn ? buf_doubles : 0;
  return 0;
}

bool spg_sync_bn_active() { return g_sync.fn != nullptr; }

static int spg_sync_allreduce(long n, hipStream_t stream) {
  const int rc = g_sync.fn(g_sync.ctx, g_sync.buf, n, (void*)stream);
  if (rc != 0) { spg_set_error("the registered BatchNorm all-reduce failed (rc %d)", rc); return -1; }
  return 0;
}

// ---- slot-synchronised BatchNorm (spg_gemm.h) ----
struct SpgSlotSync { spg_slot_allreduce_fn fn = nullptr; void* ctx = nullptr; int world = 1; };
static SpgSlotSync g_slot_sync;
extern "C" int spg_set_slot_allreduce(spg_slot_allreduce_fn fn, void* ctx, int world) {
  if (fn != nullptr) {
    SPG_CHECK_ARG(world >= 1, "slot-synchronised BatchNorm needs the world size");
    SPG_CHECK_ARG(g_sync.fn == nullptr, "slot-synchronised BatchNorm and the finalize-based mode (spg_set_bn_allreduce) exclude each other");
  }
  g_slot_sync.fn = fn; g_slot_sync.ctx = ctx; g_slot_sync.world = fn ? world : 1;
  return 0;
}
bool spg_slot_sync_active() { return g_slot_sync.fn != nullptr; }
int spg_slot_sync_world() { return g_slot_sync.world; }
static int slot_sync_now(unsigned long long* slots, size_t words, hipStream_t stream) {
  const int rc = g_slot_sync.fn(g_slot_sync.ctx, slots, (long)words, (void*)stream);
  if (rc != 0) { spg_set_error("the registered slot all-reduce failed (rc %d)", rc); return -1; }
  return 0;
}

__global__ __launch_bounds__(1024) void spg_bn_finalize_kernel(const float* __restrict__ stat, const float* __restrict__ cnt,
                                                               int nparts, long M, int N, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* running_mean,
                                                               float* running_var, float momentum, float eps,
                                                               int update_times, float* mean_o, float* rstd_o, float* s_o,
                                                               float* t_o, double* __restrict__ scratch, int* counters,
                                                               double* __restrict__ sync_out) {
  __shared__ double r0[16][17], r1[16][17], r2[16][17];
  __shared__ int s_last;
  const int cx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  const int nslices = gridDim.y, slice = blockIdx.y;
  const int per = (nparts + nslices - 1) / nslices;
  const int b0 = slice * per, b1 = min(nparts, b0 + per);
  SpgBnChannel ch = {1.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 16 && c < N && sync_out == nullptr) ch = spg_bn_channel(c, gamma, beta, running_mean, running_var);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (c < N)
    for (int b = b0 + ty; b < b1; b += 512) {                    // eight partials in flight per thread: the 512 partials of a
      float nb[8], mb[8], qb[8];                                 // persistent launch are ONE memory round trip (latency-bound loop)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = b + 64 * u;
        const bool ok = bb < b1;
        const long o = ((long)(ok ? bb : b) * 2) * N + c;
        nb[u] = ok ? cnt[bb] : 0.f;                              // rows behind this partial (written by the GEMM)
        mb[u] = stat[o];
        qb[u] = ok ? stat[o + N] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += (double)nb[u] * (double)mb[u];
        a1 += (double)nb[u] * (double)mb[u] * (double)mb[u];
        a2 += (double)qb[u];
      }
    }
  a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
  a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 16) { r0[wave][cx] = a0; r1[wave][cx] = a1; r2[wave][cx] = a2; }
  __syncthreads();
  if (threadIdx.x < 16) {
    a0 = a1 = a2 = 0.0;
    for (int k = 0; k < 16; ++k) { a0 += r0[k][cx]; a1 += r1[k][cx]; a2 += r2[k][cx]; }
  }
  if (nslices > 1) {
    if (threadIdx.x < 16 && c < N) {
      scratch[((long)slice * 3 + 0) * N + c] = a0;
      scratch[((long)slice * 3 + 1) * N + c] = a1;
      scratch[((long)slice * 3 + 2) * N + c] = a2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nslices - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-reset
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 16 && c < N) {
      a0 = a1 = a2 = 0.0;
      for (int k = 0; k < nslices; ++k) {      // fixed order: deterministic
        a0 += scratch[((long)k * 3 + 0) * N + c];
        a1 += scratch[((long)k * 3 + 1) * N + c];
        a2 += scratch[((long)k * 3 + 2) * N + c];
      }
    }
  }
  if (threadIdx.x >= 16 || c >= N) return;
  if (sync_out != nullptr) {     // synchronised BatchNorm: publish the local sums, the host all-reduces them
    sync_out[0 * (long)N + c] = a0;
    sync_out[1 * (long)N + c] = a1;
    sync_out[2 * (long)N + c] = a2;
    if (c == 0) sync_out[3 * (long)N] = (double)M;
    return;
  }
  spg_bn_finish(a0, a1, a2, (double)M, c, ch, running_mean, running_var, momentum, eps, update_times, mean_o,
                rstd_o, s_o, t_o);
}

// second half of the synchronised mode: sums over all ranks -> mean / rstd / scale / shift
__global__ void spg_bn_finish_kernel(const double* __restrict__ sync, int N, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* running_mean, float* running_var,
                                     float momentum, float eps, int update_times, float* mean_o, float* rstd_o,
                                     float* s_o, float* t_o) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  spg_bn_finish(sync[c], sync[(long)N + c], sync[2 * (long)N + c], sync[3 * (long)N], c,
                spg_bn_channel(c, gamma, beta, running_mean, running_var), running_mean, running_var, momentum, eps, update_times,
                mean_o, rstd_o, s_o, t_o);
}

size_t spg_bn_finalize_scratch_doubles(int N) { return (size_t)SPG_FIN_SLICES * 3 * N; }

int spg_launch_bn_finalize(const float* stat, const float* stat_cnt, int nparts, long M, int N, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                           int update_times, float* mean, float* rstd, float* s, float* t, double* scratch,
                           hipStream_t stream) {
  SPG_CHECK_ARG(stat != nullptr && stat_cnt != nullptr && nparts > 0, "statistics partials");
  const int gx = spg_cdiv(N, 16);
  // many partials: slice the reduction over more workgroups (one CU cannot pull megabytes of partials quickly)
  const int slice_min = g_tune[SPG_TUNE_FIN_SLICE_MIN] > 0 ? g_tune[SPG_TUNE_FIN_SLICE_MIN] : 512;     // measured (tools/tune_sweep.py): 256 / 512 equal, 1024 +30 us per step
  int slices = (scratch != nullptr && nparts > slice_min) ? SPG_FIN_SLICES : 1;
  int* counters = nullptr;
  if (slices > 1) {
    counters = spg_fin_counter_window(gx);
    if (counters == nullptr) slices = 1;
  }
  double* sync = nullptr;
  if (g_sync.fn != nullptr) {
    SPG_CHECK_ARG(3L * N + 1 <= g_sync.ndoubles, "synchronised BatchNorm buffer too small for this layer");
    sync = g_sync.buf;
  }
  hipLaunchKernelGGL(spg_bn_finalize_kernel, dim3(gx, slices), dim3(1024), 0, stream, stat, stat_cnt, nparts, M, N,
                     gamma, beta, running_mean, running_var, momentum, eps, update_times, mean, rstd, s, t, scratch, counters,
                     sync);
  SPG_LAUNCH_CHECK();
  if (sync != nullptr) {
    SPG_TRY(spg_sync_allreduce(3L * N + 1, stream));
    hipLaunchKernelGGL(spg_bn_finish_kernel, dim3(spg_cdiv(N, 256)), dim3(256), 0, stream, sync, N, gamma, beta, running_mean,
                       running_var, momentum, eps, update_times, mean, rstd, s, t);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

__global__ void spg_bn_eval_kernel(int N, const float* gamma, const float* beta, const float* rm, const float* rv,
                                   float eps, float* s, float* t) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const double rstd = 1.0 / sqrt((double)rv[c] + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
  s[c] = (float)(g * rstd);
  t[c] = (float)(be - (double)rm[c] * g * rstd);
}

// all BatchNorm layers of a network in ONE launch (eval mode: the constants depend on the parameters only)
__global__ void spg_bn_eval_batch_kernel(const SpgBnEvalBatch b, float eps) {
  const SpgBnEvalJob j = b.jobs[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= j.N) return;
  const double rstd = 1.0 / sqrt((double)j.rv[c] + (double)eps);
  const double g = j.gamma ? (double)j.gamma[c] : 1.0, be = j.beta ? (double)j.beta[c] : 0.0;
  j.s[c] = (float)(g * rstd);
  j.t[c] = (float)(be - (double)j.rm[c] * g * rstd);
}

int spg_launch_bn_eval_batch(const SpgBnEvalBatch& b, float eps, hipStream_t stream) {
  if (b.njobs == 0) return 0;
  int nmax = 1;
  for (int i = 0; i < b.njobs; ++i) nmax = b.jobs[i].N > nmax ? b.jobs[i].N : nmax;
  hipLaunchKernelGGL(spg_bn_eval_batch_kernel, dim3(spg_cdiv(nmax, 64), b.njobs), dim3(64), 0, stream, b, eps);
  SPG_LAUNCH_CHECK();
  return 0;
}

int spg_launch_bn_eval(int N, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* s, float* t, hipStream_t stream) {
  hipLaunchKernelGGL(spg_bn_eval_kernel, dim3(spg_cdiv(N, 64)), dim3(64), 0, stream, N, gamma, beta, running_mean,
                     running_var, eps, s, t);
  SPG_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(1024) void spg_bn_bwd_finalize_kernel(const float* __restrict__ stat, int ntile, int ldstat,
                                                                   long count, int N, const float* __restrict__ s,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* consts,
                                                                   float* dgamma, float* dbeta,
                                                                   double* __restrict__ scratch, int* counters,
                                                                   double* __restrict__ sync_out) {
  __shared__ double r0[16][17], r1[16][17];
  __shared__ int s_last;
  const int cx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 channels x 64 partial-groups per workgroup
  const int c = blockIdx.x * 16 + cx;
  const int nslices = gridDim.y, slice = blockIdx.y;           // sliced like spg_bn_finalize_kernel
  const int per = (ntile + nslices - 1) / nslices;
  const int t0 = slice * per, t1 = min(ntile, t0 + per);
  float ps = 0.f, pmean = 0.f, prstd = 0.f;                     // fetched now: in flight during the reduction
  if (threadIdx.x < 16 && c < N && sync_out == nullptr) { ps = s[c]; pmean = mean[c]; prstd = rstd[c]; }
  double a = 0.0, b = 0.0;
  if (c < N)
    for (int t = t0 + ty; t < t1; t += 512) {                    // eight partials in flight per thread (one round trip for 512)
      float x[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tt = t + 64 * u;
        const bool ok = tt < t1;
        const long o = ((long)(ok ? tt : t) * 2) * ldstat + c;
        x[u] = ok ? stat[o] : 0.f;
        y[u] = ok ? stat[o + ldstat] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)x[u]; b += (double)y[u]; }
    }
  a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
  a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 16) { r0[wave][cx] = a; r1[wave][cx] = b; }
  __syncthreads();
  if (threadIdx.x < 16) {
    a = b = 0.0;
    for (int k = 0; k < 16; ++k) { a += r0[k][cx]; b += r1[k][cx]; }
  }
  if (nslices > 1) {
    if (threadIdx.x < 16 && c < N) {
      scratch[((long)slice * 2 + 0) * N + c] = a;
      scratch[((long)slice * 2 + 1) * N + c] = b;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nslices - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 16 && c < N) {
      a = b = 0.0;
      for (int k = 0; k < nslices; ++k) {
        a += scratch[((long)k * 2 + 0) * N + c];
        b += scratch[((long)k * 2 + 1) * N + c];
      }
    }
  }
  if (threadIdx.x >= 16 || c >= N) return;
  if (dbeta) dbeta[c] = (float)a;      // parameter gradients stay local sums (the gradient all-reduce adds the ranks)
  if (dgamma) dgamma[c] = (float)b;
  if (sync_out != nullptr) {
    sync_out[c] = a;
    sync_out[(long)N + c] = b;
    if (c == 0) sync_out[2 * (long)N] = (double)count;
    return;
  }
  const double c1 = a / (double)count, c2 = b / (double)count;
  consts[0 * N + c] = ps;
  consts[1 * N + c] = (float)c1;
  consts[2 * N + c] = pmean;
  consts[3 * N + c] = (float)((double)ps * c2 * (double)prstd);
}

__global__ void spg_bn_bwd_finish_kernel(const double* __restrict__ sync, int N, const float* __restrict__ s,
                                         const float* __restrict__ mean, const float* __restrict__ rstd, float* consts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const double count = sync[2 * (long)N];
  const double c1 = sync[c] / count, c2 = sync[(long)N + c] / count;
  consts[0 * N + c] = s[c];
  consts[1 * N + c] = (float)c1;
  consts[2 * N + c] = mean[c];
  consts[3 * N + c] = (float)((double)s[c] * c2 * (double)rstd[c]);
}

int spg_launch_bn_bwd_finalize(const float* stat, int ntile, int ldstat, long count, int N, const float* s,
                               const float* mean, const float* rstd, float* consts, float* dgamma, float* dbeta,
                               double* scratch, hipStream_t stream) {
  const int gx = spg_cdiv(N, 16);
  const int slice_min = g_tune[SPG_TUNE_FIN_SLICE_MIN] > 0 ? g_tune[SPG_TUNE_FIN_SLICE_MIN] : 512;     // measured (tools/tune_sweep.py): 256 / 512 equal, 1024 +30 us per step
  int slices = (scratch != nullptr && ntile > slice_min) ? SPG_FIN_SLICES : 1;
  int* counters = nullptr;
  if (slices > 1) {
    counters = spg_fin_counter_window(gx);
    if (counters == nullptr) slices = 1;
  }
  double* sync = nullptr;
  if (g_sync.fn != nullptr) {
    SPG_CHECK_ARG(2L * N + 1 <= g_sync.ndoubles, "synchronised BatchNorm buffer too small for this layer");
    sync = g_sync.buf;
  }
  hipLaunchKernelGGL(spg_bn_bwd_finalize_kernel, dim3(gx, slices), dim3(1024), 0, stream, stat, ntile, ldstat, count, N, s,
                     mean, rstd, consts, dgamma, dbeta, scratch, counters, sync);
  SPG_LAUNCH_CHECK();
  if (sync != nullptr) {
    SPG_TRY(spg_sync_allreduce(2L * N + 1, stream));
    hipLaunchKernelGGL(spg_bn_bwd_finish_kernel, dim3(spg_cdiv(N, 256)), dim3(256), 0, stream, sync, N, s, mean, rstd, consts);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void spg_pool_select_kernel(const float* pmax, const float* pmin, const int* imax, const int* imin,
                                       const float* s, int G, int N, int wi, const float* extra, int nextra, float* out,
                                       long ldo, int* aidx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = N + nextra;
  if (i >= (long)G * W) return;
  const long g = i / W;
  const int c = (int)(i - g * W);
  if (c < N) {
    // combine the per-wave partials of the tile (first index wins ties), then pick max or min by the sign of the BN scale
    float vmx = -FLT_MAX, vmn = FLT_MAX;
    int imx = INT_MAX, imn = INT_MAX;
    for (int w = 0; w < wi; ++w) {
      const long o = (g * wi + w) * N + c;
      const float ov = pmax[o], pv = pmin[o];
      const int oi = imax ? imax[o] : 0, pi = imin ? imin[o] : 0;      // no indices in inference
      if (ov > vmx || (ov == vmx && oi < imx)) { vmx = ov; imx = oi; }
      if (pv < vmn || (pv == vmn && pi < imn)) { vmn = pv; imn = pi; }
    }
    const bool up = s[c] >= 0.f;
    out[g * ldo + c] = up ? vmx : vmn;
    if (aidx) aidx[g * ldo + c] = up ? imx : imn;   // aidx shares the leading dimension of `out`
  } else {
    out[g * ldo + c] = extra[g * nextra + (c - N)];
  }
}

int spg_launch_pool_select(const float* pmax, const float* pmin, const int* imax, const int* imin, const float* s,
                           int G, int N, int rows_per_tile, const float* extra, int nextra, float* out, long ldo, int* aidx,
                           hipStream_t stream) {
  const long n = (long)G * (N + nextra);
  hipLaunchKernelGGL(spg_pool_select_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, stream, pmax, pmin, imax, imin, s,
                     G, N, spg_gemm_row_waves(rows_per_tile, N), extra, nextra, out, ldo, aidx);
  SPG_LAUNCH_CHECK();
  return 0;
}

// the same with an explicit number of partials per group and optional (null) index arrays
int spg_launch_pool_select_parts(const float* pmax, const float* pmin, const int* imax, const int* imin, const float* s,
                                 int G, int N, int nparts, const float* extra, int nextra, float* out, long ldo, int* aidx,
                                 hipStream_t stream) {
  const long n = (long)G * (N + nextra);
  hipLaunchKernelGGL(spg_pool_select_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, stream, pmax, pmin, imax, imin, s,
                     G, N, nparts, extra, nextra, out, ldo, aidx);
  SPG_LAUNCH_CHECK();
  return 0;
}

// column sums in two deterministic stages: 64 columns x 16 row-groups per workgroup over a slice of the rows,
// then the slice partials are summed in a fixed order
#define SPG_COLSUM_SLICES 64
__global__ __launch_bounds__(1024) void spg_colsum_kernel(const float* __restrict__ X, long ld, long M, int N,
                                                          long rows_per_slice, float* __restrict__ part) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const long m0 = (long)blockIdx.y * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
  float s = 0.f;
  if (c < N)
    for (long m = m0 + ty; m < m1; m += 16) s += X[m * ld + c];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][tx];
    part[(long)blockIdx.y * N + c] = t;
  }
}

size_t spg_colsum_workspace_floats(int N) { return (size_t)SPG_COLSUM_SLICES * N; }

int spg_queue_colsum(SpgReduceQueue& q, const float* X, long ld, long M, int N, float* out, hipStream_t stream) {
  long rps = (M + SPG_COLSUM_SLICES - 1) / SPG_COLSUM_SLICES;
  if (rps < 16) rps = 16;
  const int slices = spg_cdiv(M, rps);
  float* part = out;
  if (slices > 1) SPG_TRY(queue_take(q, (size_t)slices * N, &part, stream));
  SpgSmallJob sj; memset(&sj, 0, sizeof(sj));
  sj.src = X; sj.dst = part; sj.a = ld; sj.b = M; sj.c = rps; sj.i = N;
  if (!(spg_group_accepts(stream) &&
        spg_group_add(SPG_JOB_COLSUM, 0, &sj, sizeof(sj), dim3(spg_cdiv(N, 64), slices), 16 * 64 * sizeof(float), 0.0, stream, 2))) {
    hipLaunchKernelGGL(spg_colsum_kernel, dim3(spg_cdiv(N, 64), slices), dim3(1024), 0, stream, X, ld, M, N, rps, part);
    SPG_LAUNCH_CHECK();
  }
  if (slices > 1) {
    SpgReduceJob& j = q.jobs[q.njobs++];
    j.partial = part; j.out = out; j.nsplit = slices; j.n = N;
  }
  return 0;
}

int spg_launch_colsum(const float* X, long ld, long M, int N, float* out, float* work, hipStream_t stream) {
  long rps = (M + SPG_COLSUM_SLICES - 1) / SPG_COLSUM_SLICES;
  if (rps < 16) rps = 16;
  const int slices = spg_cdiv(M, rps);
  hipLaunchKernelGGL(spg_colsum_kernel, dim3(spg_cdiv(N, 64), slices), dim3(1024), 0, stream, X, ld, M, N, rps,
                     slices == 1 ? out : work);
  SPG_LAUNCH_CHECK();
  if (slices > 1) {
    hipLaunchKernelGGL(spg_reduce_partials_kernel, dim3(spg_cdiv(N, 64)), dim3(1024), 0, stream, (const float*)work, slices,
                       (long)N, out);
    SPG_LAUNCH_CHECK();
  }
  return 0;
}

// dst[r, c] = c < cols ? src[r, c] : 0   (copy with a zero-padded leading dimension)
__global__ void spg_pad_rows_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long rows,
                                    int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ldd) return;
  const long r = i / ldd;
  const int c = (int)(i - r * ldd);
  dst[i] = c < cols ? src[r * lds_ + c] : 0.f;
}

int spg_launch_pad_rows(const float* src, long lds, float* dst, long ldd, long rows, int cols, hipStream_t stream) {
  const long n = rows * ldd;
  SpgSmallJob sj; memset(&sj, 0, sizeof(sj));
  sj.src = src; sj.dst = dst; sj.a = lds; sj.b = ldd; sj.c = rows; sj.i = cols;
  if (spg_group_accepts(stream) && spg_group_add(SPG_JOB_PAD_ROWS, 0, &sj, sizeof(sj), dim3(spg_cdiv(n, 256)), 0, 0.0, stream, 0)) return 0;
  hipLaunchKernelGGL(spg_pad_rows_kernel, dim3(spg_cdiv(n, 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols);
  SPG_LAUNCH_CHECK();
  return 0;
}

__global__ void spg_stn_dT_kernel(const float* __restrict__ clouds, int Ctot, int P, int G, const float* __restrict__ dxy,
                                  long ldd, float* __restrict__ dT) {
  // one wavefront per superpoint
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= G) return;
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  for (int p = lane; p < P; p += 64) {
    const float x = clouds[((long)g * Ctot + 0) * P + p], y = clouds[((long)g * Ctot + 1) * P + p];
    const float d0 = dxy[((long)g * P + p) * ldd + 0], d1 = dxy[((long)g * P + p) * ldd + 1];
    a00 = fmaf(x, d0, a00); a01 = fmaf(x, d1, a01); a10 = fmaf(y, d0, a10); a11 = fmaf(y, d1, a11);
  }
  a00 = spg_wave_sum(a00); a01 = spg_wave_sum(a01); a10 = spg_wave_sum(a10); a11 = spg_wave_sum(a11);
  if (lane == 0) {
    dT[(long)g * 4 + 0] = a00; dT[(long)g * 4 + 1] = a01; dT[(long)g * 4 + 2] = a10; dT[(long)g * 4 + 3] = a11;
  }
}

int spg_launch_stn_dT(const float* clouds, int Ctot, int P, int G, const float* dxy, long ldd, float* dT,
                      hipStream_t stream) {
  hipLaunchKernelGGL(spg_stn_dT_kernel, dim3(spg_cdiv(G, 4)), dim3(256), 0, stream, clouds, Ctot, P, G, dxy, ldd, dT);
  SPG_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// grouped launches: ONE kernel runs a list of mutually independent small jobs (spg_gemm.h: SpgGroupScope)
// ---------------------------------------------------------------------------------------------
// column sums with the summation order of spg_colsum_kernel (16 row groups per slice, combined in fixed order), 256 threads:
// a thread carries the four row groups ty, ty + 4, ty + 8, ty + 12 in separate accumulators -- bit-identical results
__device__ __forceinline__ void spg_colsum_body(const SpgSmallJob& j, const int bx, const int by, float* __restrict__ red) {
  const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const int c = bx * 64 + tx;
  const long m0 = (long)by * j.c, m1 = min(j.b, m0 + j.c);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < j.i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      for (long m = m0 + tq + 4 * u; m < m1; m += 16) s[u] += j.src[m * j.a + c];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) red[(tq + 4 * u) * 64 + tx] = s[u];
  __syncthreads();
  if (tq == 0 && c < j.i) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 64 + tx];
    j.dst[(long)by * j.i + c] = t;
  }
}

// out[i] = sum_k partial[k][i] with the summation order of spg_reduce_batch_kernel (16 split groups k = g, g + 16, ... each in
// sequence, then the groups in order): bit-identical results; 64 elements per workgroup instead of 256, i.e. 4x the
// workgroups -- as a job of a grouped launch the reduction should finish in the shadow of the launch's other jobs
__device__ __forceinline__ void spg_reduce_body(const SpgReduceJob& job, const int bx, f32x4* __restrict__ red) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long i = ((long)bx * 16 + tx) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (job.n & 3) == 0 && ((((uintptr_t)job.out) | ((uintptr_t)job.partial)) & 15) == 0;   // wave-uniform
  if (vec) {
    if (i < job.n) {
      int k = ty;
      for (; k + 48 < job.nsplit; k += 64) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(job.partial + (long)k * job.n + i);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 16) * job.n + i);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 32) * job.n + i);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(job.partial + (long)(k + 48) * job.n + i);
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; k < job.nsplit; k += 16) s += *reinterpret_cast<const f32x4*>(job.partial + (long)k * job.n + i);
    }
  } else {
    for (int e = 0; e < 4; ++e)
      if (i + e < job.n)
        for (int k = ty; k < job.nsplit; k += 16) s[e] += job.partial[(long)k * job.n + i + e];
  }
  red[ty * 16 + tx] = s;
  __syncthreads();
  if (ty == 0 && i < job.n) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + tx];      // fixed order: deterministic
    if (vec) {
      *reinterpret_cast<f32x4*>(job.out + i) = t;
    } else {
      for (int e = 0; e < 4; ++e)
        if (i + e < job.n) job.out[i + e] = t[e];
    }
  }
}

template <bool HEAVY>
__device__ __forceinline__ void spg_multi_body() {
  extern __shared__ f32x4 smem[];
  // the job table is indexed dynamically: read it where it lies -- in the kernel-argument segment (constant address space,
  // scalar loads) -- instead of through the by-value parameter, which the compiler would copy to scratch (3.9 KB per lane)
  const SpgMultiArgs& a = *(const SpgMultiArgs*)(spg_kernarg_ptr<SpgMultiArgs>)__builtin_amdgcn_kernarg_segment_ptr();
  // the job table (first_block, headers: the first 592 bytes) in one batch: the header of the workgroup's job is a dependent load
  // (-0.26 % of the step)
  spg_touch_params<(int)offsetof(SpgMultiArgs, arena)>((spg_kernarg_ptr<unsigned>)__builtin_amdgcn_kernarg_segment_ptr());
#ifdef SPG_ATTRIBUTION
  const unsigned long long trace_t0 = wall_clock64();
#endif
  // (independent loads, issued together: a dependent scan would pay one scalar-memory round trip per entry)
  int j = 0;
  const int nj = a.njobs;
#pragma unroll
  for (int k = 1; k < SPG_GROUP_MAX_JOBS; ++k) j += (k < nj && (int)blockIdx.x >= a.first_block[k]) ? 1 : 0;
  j = __builtin_amdgcn_readfirstlane(j);
  const SpgJobHdr h = a.hdr[j];
  const int b = (int)blockIdx.x - a.first_block[j];
  const int bx = b % h.gx + h.bx0, by = (b / h.gx) % h.gy, bz = b / (h.gx * h.gy);
  const unsigned char* P = a.arena + h.offset;
#define SPG_P(T) (*reinterpret_cast<const T*>(P))
  if (h.kind == SPG_JOB_GEMM) {
    spg_touch_params<(int)sizeof(SpgGemmParams)>((spg_kernarg_ptr<unsigned>)P);      // (GEMM jobs only: -0.3 % of the step; for every job kind: +1 %)
    switch (h.variant) {
      case 0: spg_rowgemm_body<32, 128, 1, 4, false, SPG_PRO_IDENT, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 1: spg_rowgemm_body<32, 128, 1, 4, false, SPG_PRO_IDENT, true>(SPG_P(SpgGemmParams), bx, by); break;
      case 2: spg_rowgemm_body<32, 128, 1, 4, false, SPG_PRO_AFFINE, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 3: spg_rowgemm_body<32, 128, 1, 4, false, SPG_PRO_AFFINE, true>(SPG_P(SpgGemmParams), bx, by); break;
      case 4: spg_rowgemm_body<32, 128, 1, 4, false, -1, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 5: spg_rowgemm_body<32, 128, 1, 4, true, SPG_PRO_IDENT, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 6: spg_rowgemm_body<32, 128, 1, 4, true, SPG_PRO_IDENT, true>(SPG_P(SpgGemmParams), bx, by); break;
      case 7: spg_rowgemm_body<32, 128, 1, 4, true, SPG_PRO_BNBWD, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 8: if constexpr (HEAVY) spg_rowgemm_body<32, 128, 1, 4, true, SPG_PRO_BNBWD, true>(SPG_P(SpgGemmParams), bx, by); break;
      case 9: spg_rowgemm_body<32, 128, 1, 4, true, -1, false>(SPG_P(SpgGemmParams), bx, by); break;
      case 10: spg_fewrow_sk_body<false, SPG_PRO_IDENT>(SPG_P(SpgGemmParams), bx, by); break;
      case 11: spg_fewrow_sk_body<false, SPG_PRO_AFFINE>(SPG_P(SpgGemmParams), bx, by); break;
      case 12: spg_fewrow_sk_body<true, SPG_PRO_IDENT>(SPG_P(SpgGemmParams), bx, by); break;
      case 13: spg_fewrow_sk_body<true, SPG_PRO_BNBWD>(SPG_P(SpgGemmParams), bx, by); break;
      default: break;
    }
  } else if (h.kind == SPG_JOB_WGRAD) {
    switch (h.variant) {
#define SPG_X(id, IT_, JT_, WI_, WJ_, AM_, BM_, FU_, CS_, LI_) \
      case id: if constexpr (HEAVY || LI_) spg_wgrad_body<IT_, JT_, WI_, WJ_, AM_, BM_, FU_, 0, CS_>(SPG_P(SpgWgradParams), bx, by, bz); break;
      SPG_WGRAD_VARIANTS(SPG_X)
#undef SPG_X
      default: break;
    }
  } else if (h.kind == SPG_JOB_COLSUM) {
    spg_colsum_body(SPG_P(SpgSmallJob), bx, by, reinterpret_cast<float*>(smem));
  } else if (h.kind == SPG_JOB_EDGE_WGRAD) {
    spg_ecc_edge_wgrad_body(SPG_P(SpgEdgeWgrad), bx);
  } else if (h.kind == SPG_JOB_PAD_ROWS) {
    const SpgSmallJob& q = SPG_P(SpgSmallJob);
    const long i = (long)bx * SPG_THREADS + threadIdx.x;
    if (i < q.c * q.b) {
      const long r = i / q.b;
      const int c = (int)(i - r * q.b);
      q.dst[i] = c < q.i ? q.src[r * q.a + c] : 0.f;
    }
  } else if (h.kind == SPG_JOB_REDUCE) {
    spg_reduce_body(SPG_P(SpgReduceJob), bx, smem);
  } else if (h.kind == SPG_JOB_ZERO) {
    const SpgSmallJob& q = SPG_P(SpgSmallJob);
    for (long i = (long)bx * 4 * SPG_THREADS + threadIdx.x; i < min(q.b, (long)(bx + 1) * 4 * SPG_THREADS); i += SPG_THREADS) q.dst[i] = 0.f;
  }
#undef SPG_P
#ifdef SPG_ATTRIBUTION
  if (a.trace != nullptr) {      // span of the job = [earliest start, latest end] over its workgroups (tools: bench.py --group-trace)
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicMin(a.trace + 2 * j, trace_t0);
      atomicMax(a.trace + 2 * j + 1, (unsigned long long)wall_clock64());
    }
  }
#endif
}
__global__ __launch_bounds__(SPG_THREADS, 2) void spg_multi_kernel(const SpgMultiArgs a_by_value) { spg_multi_body<true>(); }
__global__ __launch_bounds__(SPG_THREADS, 4) void spg_multi_light_kernel(const SpgMultiArgs a_by_value) { spg_multi_body<false>(); }

namespace {
struct SpgGroupState {
  bool open = false;
  hipStream_t st = nullptr;
  size_t lds = 0, used = 0;   // dynamic LDS of the launch (max over the jobs); bytes of the parameter arena in use
  double flops = 0.0;
  bool heavy = false;         // a job needs the 2-workgroups-per-CU build (spg_multi_kernel); else spg_multi_light_kernel
  int rc = 0;                 // first launch error since the scope was opened
  std::function<int()> direct;      // stand-alone launch of the group's FIRST job (a group of one leaves as the kernel it is)
  std::vector<std::pair<unsigned long long*, size_t>> syncs;      // slot-synchronised BatchNorm: slots this group's jobs add to
  long added = 0;             // jobs ever taken (monotonic: tells a launch function whether ITS launch became a job)
  SpgMultiArgs a;
};
thread_local SpgGroupState g_grp;

// attribution build: spans of the jobs of every grouped launch (spg_group_trace below)
#ifdef SPG_ATTRIBUTION
unsigned long long* g_trace_buf = nullptr;
int g_trace_max = 0, g_trace_n = 0;
#endif
std::vector<int> g_trace_log;      // per traced launch: njobs, heavy, then per job {kind, variant, gx, gy, gz, weight}

int group_flush() {
  SpgGroupState& g = g_grp;
  const int n = g.a.njobs;
  if (n == 0) return 0;
  g.a.trace = nullptr;
  // launch order = block order: the jobs whose workgroups run longest go first, so that the tail of the launch consists of
  // short workgroups (measured: a 32-chunk data gradient behind 424 weight-gradient workgroups made its group slower than
  // the two separate launches).  Stable insertion sort by weight, descending (<= 12 entries); then the block ranges.
  for (int i = 1; i < n; ++i) {
    const SpgJobHdr h = g.a.hdr[i];
    int k = i - 1;
    while (k >= 0 && g.a.hdr[k].weight < h.weight) { g.a.hdr[k + 1] = g.a.hdr[k]; --k; }
    g.a.hdr[k + 1] = h;
  }
  long blocks = 0;
  for (int i = 0; i < n; ++i) { g.a.first_block[i] = (int)blocks; blocks += (long)g.a.hdr[i].gx * g.a.hdr[i].gy * g.a.hdr[i].gz; }
  for (int i = n; i < SPG_GROUP_MAX_JOBS; ++i) g.a.first_block[i] = 0x7fffffff;
  int rc = 0;
  {
    ProfScope prof(g.st, g.flops, SPG_PROF_TAG(3, 32, 32, 0, 0, 0));
    // a group of ONE job gains nothing from the job table (its workgroups would pay the table's scalar loads on top of their
    // own: +1.5 us measured on a 16 us launch): it leaves as the stand-alone kernel
#ifdef SPG_ATTRIBUTION
    if (!(n == 1 && g.direct) && g_trace_buf != nullptr && g_trace_n < g_trace_max) {
      g.a.trace = g_trace_buf + (size_t)g_trace_n * 2 * SPG_GROUP_MAX_JOBS;
      ++g_trace_n;
      g_trace_log.push_back(n); g_trace_log.push_back(g.heavy ? 1 : 0);
      for (int i = 0; i < n; ++i) {
        const SpgJobHdr& h = g.a.hdr[i];
        for (int v : {h.kind, h.variant, h.gx, h.gy, h.gz, h.weight}) g_trace_log.push_back(v);
      }
    }
#endif
    if (n == 1 && g.direct) rc = g.direct();
    else if (g.heavy) hipLaunchKernelGGL(spg_multi_kernel, dim3((unsigned)blocks), dim3(SPG_THREADS), g.lds, g.st, g.a);
    else hipLaunchKernelGGL(spg_multi_light_kernel, dim3((unsigned)blocks), dim3(SPG_THREADS), g.lds, g.st, g.a);
  }
  g.a.njobs = 0; g.lds = 0; g.used = 0; g.flops = 0.0; g.heavy = false; g.direct = nullptr;
  std::vector<std::pair<unsigned long long*, size_t>> syncs;
  syncs.swap(g.syncs);
  if (rc != 0) return rc;
  SPG_LAUNCH_CHECK();
  for (auto& sy : syncs) SPG_TRY(slot_sync_now(sy.first, sy.second, g.st));      // behind the launch that produced them
  return 0;
}
}  // namespace

// Attribution builds (make ATTRIBUTION=1): from now on every grouped launch of this thread records, per job, the earliest start and
// the latest end of its workgroups (100 MHz wall clock) into buf [max_launches][16 jobs][2] (device memory, pre-filled by the caller:
// starts ~0, ends 0); buf = null switches it off.  spg_group_trace_read returns the host-side log (job kinds and grids per launch).
extern "C" int spg_group_trace(void* buf, int max_launches) {
#ifdef SPG_ATTRIBUTION
  g_trace_buf = (unsigned long long*)buf; g_trace_max = buf != nullptr ? max_launches : 0; g_trace_n = 0; g_trace_log.clear();
  return 0;
#else
  (void)buf; (void)max_launches;
  return -1;
#endif
}
extern "C" int spg_group_trace_read(int* out, int max) {
  const int n = (int)g_trace_log.size();
  for (int i = 0; i < n && i < max; ++i) out[i] = g_trace_log[i];
  return n;
}

int spg_slot_sync_after(unsigned long long* slots, size_t words, hipStream_t stream, bool deferred) {
  if (!spg_slot_sync_active() || slots == nullptr || words == 0) return 0;
  if (deferred) { g_grp.syncs.emplace_back(slots, words); return 0; }
  return slot_sync_now(slots, words, stream);
}

namespace { thread_local int g_bypass = 0; }
static long group_njobs() { return g_grp.added; }
SpgGroupBypass::SpgGroupBypass(bool on) : on_(on) { if (on_) ++g_bypass; }
SpgGroupBypass::~SpgGroupBypass() { if (on_) --g_bypass; }
static bool spg_group_accepts(hipStream_t stream) { return g_grp.open && g_grp.st == stream && g_grp.rc == 0 && g_bypass == 0; }

// Jobs that RIDE in somebody else's group (rider stages, riding reductions, leaves) while > 0: the group's own jobs -- the ones the
// next launch of the stream waits for -- get the first workgroup slots of the launch (round 5: measured with the job spans of an
// attribution build, bench.py --group-trace: the head's data gradient started 17 us into a 42 us launch, behind 581 workgroups of
// riding weight gradients)
// weight: relative duration of ONE workgroup of the job (sequential reduction chunks); decides the launch order
static bool spg_group_add(int kind, int variant, const void* params, size_t bytes, dim3 grid, size_t lds, double flops, hipStream_t stream, int weight,
                          std::function<int()> direct, int bx0) {
  SpgGroupState& g = g_grp;
  const size_t need = (bytes + 15) & ~(size_t)15;
  if (!spg_group_accepts(stream) || variant < 0 || need > SPG_GROUP_ARENA_BYTES) return false;
  const long nb = (long)grid.x * grid.y * grid.z;
  if (nb <= 0 || nb > (1L << 24)) return false;
  if (g.a.njobs == SPG_GROUP_MAX_JOBS || g.used + need > SPG_GROUP_ARENA_BYTES) {
    g.rc = group_flush();
    if (g.rc != 0) return false;
  }
  SpgJobHdr& h = g.a.hdr[g.a.njobs];
  h.kind = kind; h.variant = variant; h.gx = (int)grid.x; h.gy = (int)grid.y; h.gz = (int)grid.z;
  h.offset = (int)g.used; h.weight = weight + ((g_riding > 0 || g_tune[SPG_TUNE_NO_OWNER_FIRST]) ? 0 : SPG_OWNER_WEIGHT); h.bx0 = bx0;
  g.direct = g.a.njobs == 0 ? std::move(direct) : nullptr;
  memcpy(g.a.arena + g.used, params, bytes);
  g.used += need;
  ++g.a.njobs;
  ++g.added;
  if ((kind == SPG_JOB_GEMM && !spg_gemm_variant_light(variant)) || (kind == SPG_JOB_WGRAD && !spg_wgrad_variant_light(variant))) g.heavy = true;
  if (lds > g.lds) g.lds = lds;
  g.flops += flops;
  return true;
}

bool spg_group_add_edge_wgrad(const SpgEdgeWgrad& p, hipStream_t stream) {
  return spg_group_add(SPG_JOB_EDGE_WGRAD, 0, &p, sizeof(p), dim3(spg_cdiv(p.g.E, 4)), 0, 0.0, stream, 1);
}

int spg_group_zero(float* p, size_t n, hipStream_t stream) {
  if (p == nullptr || n == 0) return 0;
  SpgSmallJob sj; memset(&sj, 0, sizeof(sj));
  sj.dst = p; sj.b = (long)n;
  if (spg_group_add(SPG_JOB_ZERO, 0, &sj, sizeof(sj), dim3(spg_cdiv((long)n, 4 * SPG_THREADS)), 0, 0.0, stream, 0)) return 0;
  hipError_t e = hipMemsetAsync(p, 0, n * sizeof(float), stream);
  if (e != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

SpgGroupScope::SpgGroupScope(hipStream_t stream) : owner_(false) {
  SpgGroupState& g = g_grp;
  if (g.open || g_tune[SPG_TUNE_NO_GROUP]) return;      // nested scope / switched off: launches stay separate
  g.open = true; g.st = stream; g.lds = 0; g.used = 0; g.flops = 0.0; g.rc = 0; g.a.njobs = 0; g.heavy = false;
  owner_ = true;
}
// ---- riders (spg_gemm.h) ----
namespace {
thread_local std::vector<SpgStage> g_riders;
thread_local size_t g_rider_next = 0;
thread_local bool g_rider_running = false;
}  // namespace
void spg_riders_push(SpgStage stage) { g_riders.push_back(std::move(stage)); }
int spg_riders_pending() { return (int)(g_riders.size() - g_rider_next); }
void spg_riders_clear() { g_riders.clear(); g_rider_next = 0; g_rider_running = false; }
// the next stage issues its launches into the open group (what the group cannot take is launched directly: stream order keeps
// it behind the previous stage, whose launch left earlier)
static int riders_step(hipStream_t st) {
  if (g_rider_running || g_rider_next >= g_riders.size()) return 0;
  g_rider_running = true;
  SpgStage stage = std::move(g_riders[g_rider_next++]);
  SpgRidingScope riding;
  const int rc = stage(st);
  g_rider_running = false;
  if (g_rider_next >= g_riders.size()) { g_riders.clear(); g_rider_next = 0; }
  return rc;
}
int spg_riders_drain(hipStream_t stream) {
  SPG_CHECK_ARG(!g_grp.open || spg_riders_pending() == 0, "spg_riders_drain inside an open group scope (the stages depend on each other)");
  while (spg_riders_pending() > 0) {
    SpgGroupScope grp(stream);          // (inside an open scope: the stages then simply join that group one after the other --
    if (!grp.active()) {                //  NOT allowed: they depend on each other; run them ungrouped instead)
      SPG_TRY(riders_step(stream));
      continue;
    }
    SPG_TRY(grp.flush());               // flush() pulls exactly one stage
  }
  return 0;
}

// ---- leaves (spg_gemm.h) ----
namespace {
struct SpgLeaf { SpgStage issue; double cost; };
thread_local std::vector<SpgLeaf> g_leaves;
thread_local size_t g_leaf_next = 0;
}  // namespace
void spg_leaf_push(SpgStage issue, double cost) { g_leaves.push_back(SpgLeaf{std::move(issue), cost}); }
int spg_leaf_pending() { return (int)(g_leaves.size() - g_leaf_next); }
void spg_leaf_clear() { g_leaves.clear(); g_leaf_next = 0; }
int spg_leaf_ride(hipStream_t stream, int launches_left) {
  if (spg_leaf_pending() == 0 || !spg_group_accepts(stream)) return 0;
  double total = 0.0;
  for (size_t i = g_leaf_next; i < g_leaves.size(); ++i) total += g_leaves[i].cost;
  const double share = total / (launches_left > 1 ? launches_left : 1);
  double taken = 0.0;
  SpgRidingScope riding;
  while (g_leaf_next < g_leaves.size() && (taken == 0.0 || taken + 0.5 * g_leaves[g_leaf_next].cost <= share)) {
    SpgLeaf leaf = std::move(g_leaves[g_leaf_next++]);
    taken += leaf.cost;
    SPG_TRY(leaf.issue(stream));
  }
  if (g_leaf_next >= g_leaves.size()) spg_leaf_clear();
  return 0;
}
int spg_leaf_drain(hipStream_t stream) {
  while (spg_leaf_pending() > 0) {
    SpgGroupScope grp(stream);      // (a scope of its own when none is open; inside an open one the leaves join it)
    while (spg_leaf_pending() > 0) {
      SpgLeaf leaf = std::move(g_leaves[g_leaf_next++]);
      SPG_TRY(leaf.issue(stream));
    }
    if (grp.active()) SPG_TRY(grp.flush());
  }
  spg_leaf_clear();
  return 0;
}

// the weight gradient `p` -> dW as `nslice` leaves (splits of its plan in contiguous ranges); the LAST leaf queues the summation
// of the partials in `q`, which must therefore outlive the leaves (spg_leaf_drain before its flush).  false: this launch has no
// grouped body / a single split / no room -- the caller issues it the ordinary way.
bool spg_queue_wgrad_leaf(SpgReduceQueue& q, SpgWgradParams p, float* dW, int nslice, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || nslice < 1 || g_tune[SPG_TUNE_NO_GROUP] || !g_tune[SPG_TUNE_LEAVES]) return false;
  int it, jt, ns, rps;
  wgrad_plan(p.M, p.N, p.K, &it, &jt, &ns, &rps);
  if (ns < 2) return false;
  p.colsum = nullptr;
  g_wgrad_probe = true; g_wgrad_probe_variant = -1;
  const int prc = spg_launch_wgrad_partials(p, dW, stream);
  g_wgrad_probe = false;
  if (prc != 0 || g_wgrad_probe_variant < 0) return false;
  if (q.njobs + 1 > SPG_MAX_REDUCE_JOBS || q.arena == nullptr || q.used + (size_t)ns * p.N * p.K > q.arena_floats) return false;
  float* part = nullptr;
  if (queue_take(q, (size_t)ns * p.N * p.K, &part, stream) != 0) return false;
  if (nslice > ns) nslice = ns;
  SpgReduceQueue* qp = &q;
  const double cost = 2.0 * (double)p.M * p.N * p.K / nslice;
  for (int sidx = 0; sidx < nslice; ++sidx) {
    const int s0 = (int)((long)ns * sidx / nslice), s1 = (int)((long)ns * (sidx + 1) / nslice);
    const bool last = sidx + 1 == nslice;
    spg_leaf_push([p, part, dW, s0, s1, ns, last, qp](hipStream_t st) -> int {
      int rc = 0;
      {
        SpgGroupScope own(st);      // no-op inside the caller's open scope; a launch of its own otherwise (drain without a scope)
        g_wgrad_slice0 = s0; g_wgrad_slice_n = s1 - s0;
        rc = spg_launch_wgrad_partials(p, part, st);
        g_wgrad_slice0 = 0; g_wgrad_slice_n = 0;
        if (rc == 0 && own.active()) rc = own.flush();
      }
      if (rc == 0 && last) {
        if (qp->njobs == SPG_MAX_REDUCE_JOBS) { spg_set_error("reduction queue full behind a weight-gradient leaf"); return 1; }
        SpgReduceJob& j = qp->jobs[qp->njobs++];
        j.partial = part; j.out = dW; j.nsplit = ns; j.n = p.N * p.K;
      }
      return rc;
    }, cost);
  }
  return true;
}

int SpgGroupScope::flush() {
  if (!owner_) return 0;
  SpgGroupState& g = g_grp;
  int rc = g.rc;
  if (rc == 0 && !g_rider_running) rc = riders_step(g.st);      // one link of a rider chain leaves with this launch
  if (rc == 0) rc = g.rc;
  if (rc == 0) rc = group_flush();
  g.rc = 0;
  return rc;
}
SpgGroupScope::~SpgGroupScope() {
  if (!owner_) return;
  // what is still collected leaves; NO rider stage is pulled here (a scope that ends is not a launch site of its owner: a stage
  // pulled now would leave alone, in front of the launches it was meant to travel with)
  if (g_grp.rc == 0) (void)group_flush();
  g_grp.rc = 0;
  g_grp.open = false;
}
