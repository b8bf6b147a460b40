// Internal (C++) interface of the fused row-GEMM kernels (spg_gemm.hip); used by the PointNet and
// filter-network orchestration code.  Not part of the C ABI.
#pragma once
#include "spg_common.h"

enum { SPG_EPI_FWD = 0, SPG_EPI_BWD = 1 };

// ---- BatchNorm statistics WITHOUT a finalize launch (round 3) ----------------------------------------------------------
// Every small dependent launch of the step costs >= ~4.5 us whatever it computes, and a train-mode BatchNorm layer used to
// cost two of them (finalize in the forward, finalize in the backward).  Instead the PRODUCER GEMM adds its per-workgroup
// (sum x, sum x^2) as exactly associative 64-bit FIXED-POINT integers (fire-and-forget agent-scope atomics, SPG_FOLD_SLOTS slots per
// channel so that at most ~64 workgroups meet on one address), and the CONSUMER GEMM -- the next layer, which needs the scale /
// shift anyway -- sums the slots of a channel in its prologue, computes mean / rstd / scale / shift in float64 and writes
// them (every workgroup writes the same bits; workgroup 0 also advances the running statistics).  Integer addition is
// order-independent, so the result is deterministic (bit-identical from run to run) although the arrival order is not.
// Representation of a double v: hi = floor(v * 2^SH), lo = frac(v * 2^SH) * 2^44 (both int64), up to 2^19 contributions per
// slot (4 M per layer) without overflow -- the launcher falls back to the finalize path beyond that.  Forward sums use
// SH = -8 (|v| <= 2^52 per contribution, quantum 2^-36: a pre-BatchNorm rms of ~3e6 over a workgroup's 512 rows is still in
// range), backward sums SH = +8 (|v| <= 2^36, quantum 2^-52: gradients are small numbers); spg_gemm.hip: spg_fx_split.  A non-finite contribution raises the flag word behind
// the slots and the consumer then produces NaN statistics, as the arithmetic it replaces would.
// Slot layout of one layer: int64 [SPG_FOLD_SLOTS slots][4 limbs: sum x hi, lo, sum x^2 hi, lo][C channels], then one flag word.
// Measured (tools/probe/bn_atomic_probe.hip): +0.3..0.6 us on the producer's tail at C = 64..256, ~1 us of consumer prologue.
#ifndef SPG_FOLD_SLOTS      // (a power of two; -DSPG_FOLD_SLOTS=n builds a variant for A/Bs)
// 4 since the end of round 6 (8 before): every consumer workgroup reads ALL slots of its channels in its prologue -- 16 instead of 32
// loads per channel -- and the producers' atomics do not contend more at 4 (same box, interleaved: 8 slots 1.116 ms per step, 4 slots
// 1.106, 2 slots 1.137).  The sums are exact integers: the slot count changes no bit of any result (tools/bitcheck.py).
#define SPG_FOLD_SLOTS 4
#endif
#define SPG_FOLD_MAX_CONTRIBUTIONS ((long)SPG_FOLD_SLOTS << 17)      // (a quarter of the representable count: SPG_FOLD_SLOTS x 2^19)
inline size_t spg_fold_slot_words(int C) { return (size_t)SPG_FOLD_SLOTS * 4 * C + 8; }
struct SpgBnFold {
  const unsigned long long* slots;   // null: nothing to do
  int C, update_times;
  float momentum, eps;
  double count;                      // rows behind the statistics (used when the slots carry no row count: older producers)
  const float *gamma, *beta;
  float *rm, *rv;                    // running statistics (may be null)
  float *mean, *rstd, *s, *t;        // outputs [C]
};
// backward: (sum dz, sum dz * xhat) of a layer -> the constants of the BatchNorm-backward prologue, finished by the first
// launch that applies them (the layer's weight gradient); same slots layout
struct SpgBnFoldBwd {
  const unsigned long long* slots;   // null: nothing to do
  int C;
  double count;
  const float *s, *mean, *rstd;      // forward constants of the layer [C]
  float* consts;                     // out [4][C] = {s, c1, mean, s * c2 * rstd}
  float *dgamma, *dbeta;             // out [C] (may be null)
  double grad_mul;                   // slot-synchronised BatchNorm: the sums are those of ALL ranks and every rank writes dgamma / dbeta,
                                     // which the gradient all-reduce then adds up -- each writes sum * (1 / ranks) (0 = 1: single rank)
};
// ---- slot-synchronised BatchNorm (round 5): data-parallel ranks normalise over the union of their batches by ALL-REDUCING THE
// FIXED-POINT SLOTS THEMSELVES (int64 sums: exact and order-independent -- the synchronised statistics are bit-identical on every
// rank and for every rank count) between the producer launch and the consumer launch.  Everything built on the slots keeps
// working: the folds, the fused convolution backward, the one-pass first layers, grouped launches, riders, spg_train_step.  The
// launch functions issue the collective themselves behind every launch that produced slots (behind the group's launch for a job
// of a grouped launch).  The consumers' row counts travel WITH the sums: every producer launch adds its rows to the spare word
// behind the slots' flag word (spg_fold.h: spg_slots_count_add), so the all-reduced slots carry the rows of all ranks.
bool spg_slot_sync_active();
int spg_slot_sync_world();
// to be called behind a launch that added to `slots` (words int64): immediately, or behind the open group's launch (deferred)
int spg_slot_sync_after(unsigned long long* slots, size_t words, hipStream_t stream, bool deferred);
#define SPG_FC_ROWS 32   // rows per workgroup for the few-row GEMMs (FC layers over superpoints, filter net over edges)

// Y[M,N] = prologue(A)[M,K] @ W[N,K]^T (+ bias), with a fused epilogue.
struct SpgGemmParams {
  SpgOperand a;
  const float* W;     // [N, K] row-major (w_red == 0) or [K, N] row-major (w_red == 1: the untransposed weight of a dgrad)
  long ldw;
  int w_red;
  const float* bias;  // [N] or null
  int M, N, K;
  int rows_per_tile;  // <= 128: rows handled by one workgroup (points per superpoint for the 1x1 convs)
  int epi;
  float* Y;           // [M, ldy] or null (pool-only forward)
  long ldy;
  // SPG_EPI_FWD: per-tile BatchNorm partials (mean, M2) and per-tile max/min pooling of the raw output
  float* stat;        // [parts][2][N] or null: per-wave partials (forward: mean, M2; backward: sum dz, sum dz*xhat).  The number
                      // of parts is decided by the launcher (one per tile and row-wave, or one per persistent workgroup and
                      // row-wave) and returned through spg_launch_gemm's stat_parts
  float* stat_cnt;    // forward: [parts] rows behind every partial (required with stat)
  unsigned long long* stat_slots;   // forward, instead of stat / stat_cnt: fixed-point slots of this layer (SpgBnFold above)
  long stat_rows;     // rows this launch stands for in the statistics' row count (spg_fold.h: spg_slots_count_add); 0 = M.  (The data
                      // gradient of the FC layer behind a max-pool has M = superpoints but feeds the statistics of the pooled layer over
                      // ALL points.)
  SpgBnFold fold;     // forward: statistics of the layer that PRODUCED operand `a`, to be finished in this launch's prologue
  SpgBnFoldBwd fold_bwd;   // data gradient: BatchNorm-backward sums of the `a` operand's layer, finished in this launch's prologue
                           // too when the layer's weight gradient (which otherwise does it first) runs in the SAME grouped launch
  // max-pool over the rows of a tile (= the points of a superpoint), fused: BatchNorm is monotone per channel, so the
  // pooled normalised value is the raw MAX where the BatchNorm scale is >= 0 and the raw MIN otherwise, and the sign of
  // the scale gamma * rstd is the sign of gamma -- known before the statistics are.  pool_out [ntile, pool_ld] receives
  // the selected raw extremum per (tile, channel), pool_idx the row inside the tile that holds it (first one on ties,
  // like max_pool1d), columns N .. N + pool_nextra - 1 the concatenated global features.
  float* pool_out;    // null: no pooling
  int* pool_idx;      // may be null (inference)
  const float* pool_sign;   // [N] BatchNorm weight (null: all >= 0)
  const float* pool_extra;  // [ntile, pool_nextra] or null
  long pool_ld;
  int pool_nextra;
  // SPG_EPI_BWD: columns are the channels of the producer layer; ReLU mask and BatchNorm-backward sums
  const float* Yp;    // raw output of the producer layer [M, ldyp] (null: no mask / no stats)
  long ldyp;
  const float* ms;    // producer BN scale / shift (null: 1 / 0)
  const float* mt;
  int vec_store;     // set by the launcher (full tiles, aligned output rows): dwordx4 stores through LDS
  int mask_relu;
  int n_mask;         // columns >= n_mask are passed through (concatenated global features)
  const float* mmean; // producer BN batch mean / rstd (null: no stats)
  const float* mrstd;
  // set by the launcher (full-tile fast path): work-item map of the persistent chunk stream -- a workgroup handles row
  // tiles tile0, tile0 + rstride, ... (< ntile) of one column tile; remap: XCD-aware block -> (row tile, column tile) map
  int remap, ncol, ntile, rstride;
  int stat_accum;     // persistent launches: one statistics partial per workgroup and row-wave (accumulated over its tiles)
  int dbg;            // timing-attribution switches (spg_tune key 3), persistent launches only
  // opt-in bf16 / split-bf16 MFMA (spg_common.h): the weights of THIS GEMM's orientation pre-split by
  // spg_launch_split_weights -- Wb [2][N][ldwb] bf16, reduction index contiguous (hi matrix, then lo matrix); null: fp32 MFMA
  const void* Wb;
  long ldwb, wb_part_bytes;
};

// dW[N,K] = sum_m prologue_a(dY)[m, n] * prologue_b(A)[m, k]
struct SpgWgradParams {
  SpgOperand a;       // channels -> rows of dW
  SpgOperand b;       // channels -> columns of dW
  int M, N, K;
  int rows_per_split; // multiple of 32
  float* partial;     // [nsplit][N][K]
  float* colsum;      // [nsplit][N] or null: column sums of the (finished) `a` operand over each split's rows -- the bias
                      // gradient of a layer without BatchNorm comes with the weight gradient instead of from its own launch
                      // (identity `a` operands only)
  SpgBnFoldBwd fold;  // BatchNorm-backward sums of the `a` operand's layer, to be finished in this launch's prologue (or slots == null)
  int allow_lowp;     // 1: this launch may use the opt-in bf16 / split-bf16 MFMA mode (spg_tune key 7).  Set by the PointNet
                      // convolutions only; 0 (memset default) keeps fp32 MFMA whatever the shape (filter net, RNN cell, dense layer)
};

// bf16 copies of a weight matrix W [N, K] (row stride ldw floats) for the bf16 MFMA modes: fwd [2][N][ldk] (hi, lo; the
// forward GEMM's orientation) and, when `bwd` is given, the transpose [2][K][ldn] (the data gradient's orientation);
// ldk = roundup8(K), ldn = roundup8(N); all jobs of a network in one launch
struct SpgSplitJob { const float* W; long ldw; int N, K; void* fwd; void* bwd; };
#define SPG_SPLIT_MAX_JOBS 40
struct SpgSplitBatch { SpgSplitJob jobs[SPG_SPLIT_MAX_JOBS]; int njobs = 0; };
inline long spg_split_ld(int k) { return (k + 7) & ~7; }
inline size_t spg_split_bytes(int rows, int cols) { return (size_t)2 * rows * spg_split_ld(cols) * 2; }
int spg_launch_split_weights(const SpgSplitBatch& b, hipStream_t stream);
int spg_gemm_precision();      // spg_tune key 7: 0 fp32 MFMA, 1 bf16, 3 split-bf16

// ---- grouped launches (round 4) ------------------------------------------------------------------------------------------
// The step is a chain of ~80 dependent launches of which ~45 are few-row GEMMs / small reductions that take 5-15 us each
// whatever they compute (dispatch, ramp, first loads, end-of-kernel release) on a fraction of the chip.  Wherever two or
// more of them do NOT depend on each other -- a layer's data gradient and its weight gradient, the recurrent cell's three
// parameter gradients and the per-edge filter gradient, ... -- they leave as ONE launch: spg_multi_kernel runs a list of
// heterogeneous jobs, every workgroup executing the unchanged body of the kernel it replaces (same arithmetic, same
// summation order: results are bit-identical to the separate launches).  No spin waits, no cross-workgroup dependencies.
// Usage: open a scope on the stream, issue the launches (the launch functions below divert what the group can take and
// launch everything else directly -- the caller asserts that ALL launches inside one scope are mutually independent), then
// flush() (also at scope end).  One scope per thread at a time; spg_tune key 11 = 1 switches grouping off (A/B, tests).
struct SpgGroupScope {
  explicit SpgGroupScope(hipStream_t stream);
  ~SpgGroupScope();
  int flush();              // launches the collected jobs (if any) as one kernel; the scope stays open for the next group
  bool active() const { return owner_; }      // false: nested scope or grouping switched off -- launches stay separate, in order
  SpgGroupScope(const SpgGroupScope&) = delete;
  SpgGroupScope& operator=(const SpgGroupScope&) = delete;
 private:
  bool owner_;
};
// while alive (and `on`): launches of this thread are NOT diverted into the open group -- for a launch that is followed by a
// dependent plain launch inside the same stage (a BatchNorm finalize behind its GEMM)
struct SpgGroupBypass {
  explicit SpgGroupBypass(bool on = true);
  ~SpgGroupBypass();
 private:
  bool on_;
};
// zero `n` floats as a job of the open group (else hipMemsetAsync)
int spg_group_zero(float* p, size_t n, hipStream_t stream);

// ---- riders: a dependent chain of small launches that travels NEXT TO the caller's launches ----------------------------------
// The filter-generating network's forward (4 dependent few-row GEMMs) needs nothing from PointNet, and the tail of the RNN-ECC
// backward (cell / filter-network parameter gradients: 6 dependent groups) is needed by nobody before the optimiser step --
// yet both sit in the one stream, in front of resp. behind ~600 us of PointNet work, as ~45 + ~120 us of latency-bound
// launches.  A stage is a closure that issues the launches of one link of such a chain; the chain is registered with
// spg_riders_push and from then on EVERY grouped launch of this thread on that stream (SpgGroupScope::flush) first lets the
// next stage add its jobs to the group that is about to leave: stage k+1 leaves with a later launch than stage k, so the
// chain's own order is kept, and it costs no launch of its own.  spg_riders_drain runs what is left (one grouped launch per
// stage) -- the owner of the chain MUST call it before anything consumes the chain's results and before its buffers go away.
// The chain lives inside ONE C-ABI call (spg_train_step): nothing is deferred across calls.
#include <functional>
typedef std::function<int(hipStream_t)> SpgStage;
void spg_riders_push(SpgStage stage);
int spg_riders_pending();
int spg_riders_drain(hipStream_t stream);
void spg_riders_clear();

// ---- leaves (round 5): work nobody waits for before the optimiser step, issued in SLICES next to later latency-bound launches ----
// A rider chain (above) keeps an order; a leaf does not: it is a closure that issues independent jobs into the group that is open
// when it runs.  PointNet's backward pushes the weight gradient of the pooled convolution (128 -> 256: 8.4 GFLOP, 88 us as a
// launch of its own in front of the data gradient everything else waits for) and of the first convolution as leaves -- split by
// row ranges of their split plan -- and the STN head's three grouped launches (~55 us of dependent few-row links on 32
// workgroups) each take a share: spg_leaf_ride(stream, launches_left) issues ~1 / launches_left of the pending cost.  The owner
// drains what is left before anything consumes the results.  Leaves never outlive the C call that pushed them.
void spg_leaf_push(SpgStage issue, double cost);
int spg_leaf_pending();
int spg_leaf_ride(hipStream_t stream, int launches_left);
int spg_leaf_drain(hipStream_t stream);
void spg_leaf_clear();

// hipEvent bracket of one MFMA launch for bench.py's instrumented pass (spg_prof_enable; nothing happens otherwise): the launch
// counts with `flops` algorithmic FLOP under `tag` (spg_prof_tag); N, K label its row of the per-shape table
struct SpgProfSpan {
  SpgProfSpan(hipStream_t stream, double flops, int tag, int N, int K);
  ~SpgProfSpan();
  SpgProfSpan(const SpgProfSpan&) = delete;
  SpgProfSpan& operator=(const SpgProfSpan&) = delete;
 private:
  void* impl_;
};

int spg_gemm_ntiles(const SpgGemmParams& p);
// Data-gradient launches with 128-column tiles keep one workgroup per tile (no persistent stream: registers), i.e. one
// statistics contribution per tile and row-wave -- 2000 per channel on the unit scene.  That many atomics cost more on the
// kernel's tail (+11 us measured) than the finalize launch they would save (6 us): such launches keep the partials path.
// (not with slot-synchronised BatchNorm: partials + a finalize launch would normalise over this rank's rows only)
inline bool spg_gemm_bwd_stats_want_partials(const SpgGemmParams& g) { return g.w_red && g.N > 64 && g.rows_per_tile > SPG_FC_ROWS && !spg_slot_sync_active(); }
// number of statistics / pooling partials per row tile (= waves along the rows of the tile shape used for this problem)
int spg_gemm_row_waves(int rows_per_tile, int N);
int spg_launch_gemm(const SpgGemmParams& p, hipStream_t stream, int* stat_parts = nullptr);

// workspace (floats) needed by spg_launch_wgrad for a problem of this size
size_t spg_wgrad_workspace_floats(long M, int N, int K);
// floats of the per-split column sums when the bias gradient rides along (spg_queue_wgrad with db)
size_t spg_wgrad_colsum_floats(long M, int N, int K);
// dW (dense [N,K], ld = K) = reduction; `work` holds the split partials
int spg_launch_wgrad(SpgWgradParams p, float* dW, float* work, hipStream_t stream);

// Deferred reductions: the split partials of many weight gradients / column sums are summed by ONE launch at the end
// of a backward pass instead of one tiny launch each (the step is launch-bound: ~200 dependent kernels).
#define SPG_MAX_REDUCE_JOBS 40
struct SpgReduceJob {
  const float* partial;   // [nsplit][n]
  float* out;             // [n]
  int nsplit, n;
};
struct SpgReduceQueue {
  SpgReduceJob jobs[SPG_MAX_REDUCE_JOBS];
  int njobs = 0;
  float* arena = nullptr;    // scratch for all partials of one backward pass
  size_t arena_floats = 0, used = 0;
};
// db (optional): also the column sums of the `a` operand (= bias gradient), from the same launch
int spg_queue_wgrad(SpgReduceQueue& q, SpgWgradParams p, float* dW, hipStream_t stream, float* db = nullptr);
// the same weight gradient as `nslice` LEAVES (above): nothing is launched now; false = not possible for this launch (no grouped
// body for its shape, a precision mode, grouping switched off): the caller then uses spg_queue_wgrad.  `q` must outlive the leaves.
bool spg_queue_wgrad_leaf(SpgReduceQueue& q, SpgWgradParams p, float* dW, int nslice, hipStream_t stream);
// data gradient + weight gradient of a 64-input-channel convolution from ONE pass over dz (spg_gemm.hip: spg_bwdpair_kernel);
// g as for spg_launch_gemm (the data-gradient problem, statistics into slots, fold_bwd = the layer's pending sums), b = the
// layer's input operand, dW [g.K, 64] through the queue's batched reduction
bool spg_bwdpair_supported(const SpgGemmParams& g, const SpgOperand& b);
int spg_queue_bwdpair(SpgReduceQueue& q, SpgGemmParams g, const SpgOperand& b, float* dW, hipStream_t stream);
int spg_queue_partials(SpgReduceQueue& q, int nsplit, int n, float* out, float** partial, hipStream_t stream);
int spg_queue_colsum(SpgReduceQueue& q, const float* X, long ld, long M, int N, float* out, hipStream_t stream);
int spg_flush_reduce(SpgReduceQueue& q, hipStream_t stream);
// jobs whose partials are complete (in stream order) but whose summation may wait for the next batched reduction of this
// thread: spg_flush_reduce takes them along; spg_flush_deferred_reduce sums what is still waiting (no-op when nothing is)
void spg_reduce_defer(const SpgReduceJob& job);
void spg_reduce_deferred_clear();
int spg_flush_deferred_reduce(hipStream_t stream);
// hands the queue's jobs to the open group of this thread (they leave with its launch, next to its other jobs)
int spg_reduce_ride(SpgReduceQueue& q, hipStream_t stream, size_t max_bytes = (size_t)36 << 20);

// BatchNorm forward statistics: partials [nparts][2][N] (+ rows per partial, stat_cnt [nparts]) -> mean, rstd, scale s = gamma*rstd, shift t = beta - mean*s;
// running stats updated `update_times` times (run_full_monger re-runs the forward, learning/pointnet.py:167,173)
// scratch: >= spg_bn_finalize_scratch_doubles(N) doubles (or null: single-slice reduction)
size_t spg_bn_finalize_scratch_doubles(int N);
int spg_launch_bn_finalize(const float* stat, const float* stat_cnt, int nparts, long M, int N, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                           int update_times, float* mean, float* rstd, float* s, float* t, double* scratch,
                           hipStream_t stream);
// true while a BatchNorm all-reduce is registered (spg_set_bn_allreduce): every rank must then run the same sequence of
// train-mode BatchNorm layers
bool spg_sync_bn_active();
// eval mode: s, t from the running statistics; the batch form handles every BatchNorm layer of a network in one launch
struct SpgBnEvalJob { int N; const float *gamma, *beta, *rm, *rv; float *s, *t; };
#define SPG_BN_EVAL_MAX_JOBS 40
struct SpgBnEvalBatch { SpgBnEvalJob jobs[SPG_BN_EVAL_MAX_JOBS]; int njobs = 0; };
int spg_launch_bn_eval_batch(const SpgBnEvalBatch& b, float eps, hipStream_t stream);
int spg_launch_bn_eval(int N, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* s, float* t, hipStream_t stream);
// BatchNorm backward: partials [ntile][2][N] (sum dz, sum dz*xhat) -> consts[4][N] = {s, c1, mean, s*c2*rstd},
// dgamma, dbeta
int spg_launch_bn_bwd_finalize(const float* stat, int ntile, int ldstat, long count, int N, const float* s,
                               const float* mean, const float* rstd, float* consts, float* dgamma, float* dbeta,
                               double* scratch, hipStream_t stream);
// max-pool selection after the BN statistics are known: out[g, c] = s[c] >= 0 ? pmax : pmin  (+ argidx),
// out[g, N + e] = extra[g, e]; aidx uses the same leading dimension ldo as out
int spg_launch_pool_select(const float* pmax, const float* pmin, const int* imax, const int* imin, const float* s,
                           int G, int N, int rows_per_tile, const float* extra, int nextra, float* out, long ldo, int* aidx,
                           hipStream_t stream);
// work: >= spg_colsum_workspace_floats(N) floats (spg_wgrad_workspace_floats(., N, .) is always large enough)
size_t spg_colsum_workspace_floats(int N);
int spg_launch_colsum(const float* X, long ld, long M, int N, float* out, float* work, hipStream_t stream);
// dst [rows, ldd] = src [rows, cols] with zero padding (ldd >= cols): 16-byte aligned weight rows for the vector path
int spg_launch_pad_rows(const float* src, long lds, float* dst, long ldd, long rows, int cols, hipStream_t stream);
// dT[g, 2a+b] = sum_p clouds[g, a, p] * dxy[g*P + p, b]   (gradient of the 2x2 STN transform, pointnet.py:123)
int spg_launch_stn_dT(const float* clouds, int Ctot, int P, int G, const float* dxy, long ldd, float* dT,
                      hipStream_t stream);


// ---- inference-mode convolution stack of one PointNet segment in a single kernel (spg_convstack.hip) ----
#define SPG_CONVSTACK_MAX_LAYERS 8
struct SpgConvStackParams {
  const float* clouds;      // [B, Ctot, P] channel-major
  const float* stnT;        // [B, 4] or null
  int B, P, Ctot, nlayers;
  int cin[SPG_CONVSTACK_MAX_LAYERS], cout[SPG_CONVSTACK_MAX_LAYERS];
  const float *W[SPG_CONVSTACK_MAX_LAYERS], *bias[SPG_CONVSTACK_MAX_LAYERS];
  f32x4* Wp[SPG_CONVSTACK_MAX_LAYERS];   // scratch for the packed weights: spg_conv_stack_packed_floats(cin, cout) floats each
  const float *s[SPG_CONVSTACK_MAX_LAYERS], *t[SPG_CONVSTACK_MAX_LAYERS];   // eval-mode BatchNorm scale / shift (spg_launch_bn_eval*)
  float *pmax, *pmin;       // [B][4][cout_last] per-wave raw max / min of the last layer (spg_launch_pool_select, 4 partials)
};
int spg_launch_pool_select_parts(const float* pmax, const float* pmin, const int* imax, const int* imin, const float* s,
                                 int G, int N, int nparts, const float* extra, int nextra, float* out, long ldo, int* aidx,
                                 hipStream_t stream);
size_t spg_conv_stack_packed_floats(int cin, int cout);
bool spg_conv_stack_eval_supported(const SpgConvStackParams& p);
int spg_launch_conv_stack_eval(const SpgConvStackParams& p, hipStream_t stream);
