// Internal (C++) interface of the edge-conditioned-convolution / GRU kernels (spg_ecc.hip).
#pragma once
#include "spg_common.h"

// Device-side graph structure derived from GraphConvInfo's (idxn, degs)
// (reference: learning/ecc/GraphConvInfo.py:33-69): CSR by target (edges are already sorted by
// target) and the reverse CSR by source used by the atomic-free backward.
struct SpgGraph {
  int N, E;
  const int* hdr;         // [4] device header {N, Ns (number of source rows >= N), E, error flag}
  const int* rowptr;      // [N+1] start of each destination node's in-edge segment
  const int* src;         // [E]   source node of each edge (= idxn)
  const int* dst;         // [E]   destination node of each edge
  const int* rev_rowptr;  // [Ns+1] start of each source node's out-edge list
  const int* rev_eid;     // [E]   edge ids grouped by source, increasing inside a group
  const float* invdeg;    // [N]   1/in-degree, 0 for isolated nodes
};

size_t spg_graph_bytes(int N, int Ns, int E);
SpgGraph spg_graph_view(const void* workspace, int N, int E);
int spg_graph_build_impl(const int64_t* idxn, const int64_t* degs, int N, int Ns, int E, void* workspace, hipStream_t stream);

#define SPG_CELL_GRU 0
#define SPG_CELL_LSTM 1

// GRUCellEx (learning/modules.py:205-259) / LSTMCellEx (:262-316), hidden = input = 32; G = 96 (GRU) or 128 (LSTM)
struct SpgGruParams {
  const float* w_ih;    // [G,32]
  const float* w_hh;    // [G,32]
  const float* b_ih;    // [G]
  const float* b_hh;    // [G]
  const float* w_ig;    // [32,32]
  const float* b_ig;    // [32]
  int layernorm, ingate;
};

// one RNN-ECC iteration, forward: agg = mean_{in-edges} x_src (.) W_e ; h' = GRU(agg, h)
struct SpgEccStepFwd {
  SpgGraph g;
  const float* W;       // [E,32,32] (matrix) or [E,32] (vector)
  int matrix;
  const float* hin;     // h^r   rows of ld floats
  float* hout;          // h^r+1
  long ld;
  const float* agg_in;  // non-null: skip the aggregation (stand-alone GRU cell), [N, ldagg]
  float* agg_save;      // [N, ldagg] or null
  long ldagg;
  int do_gru;           // 0: aggregation only (result in agg_save)
  SpgGruParams gru;
  int cell;             // SPG_CELL_GRU / SPG_CELL_LSTM
  const float* cin;     // LSTM: cell state c^r (rows of ld floats) or null (= zeros)
  float* cout;          // LSTM: c^r+1
};
int spg_launch_ecc_step_fwd(const SpgEccStepFwd& p, hipStream_t stream);

struct SpgEccStepBwd {
  SpgGraph g;
  const float* W;
  int matrix;
  // phase 1: dH = dcat + dhdir + sum_{out-edges} W_e . Gnext[dst]
  const float* dcat;    // [N, ldc] slice (already offset to this state's columns) or null
  long ldc;
  const float* Gnext;   // [N, ldg] gradient wrt the aggregate of the NEXT iteration (already / deg), or null
  long ldg;
  float* dhdir;         // [N,32] in/out: direct GRU-path gradient wrt the hidden state
  int use_dhdir;
  float* gx;            // final_only: output gradient wrt h^0 [N,32]
  int final_only;
  // phase 2: GRU backward of this iteration
  const float* hin;     // h^r
  long ld;
  const float* agg;     // aggregate (GRU input before the input gate) of this iteration [N, ldagg]
  long ldagg;
  float* Gcur;          // [N, ldg] out: d agg / deg
  float* dgi;           // [N, ld96] outputs for the deferred weight gradients
  float* dgh;
  float* dui;
  float* duh;
  long ld96;
  float* dpre;          // [N, ld32]
  float* xg;            // [N, ld32]
  long ld32;
  SpgGruParams gru;
  // LSTM only (dui / duh unused: the biases sit in front of the row normalisation, their gradients are the column
  // sums of dgi / dgh; ld96 is then the leading dimension of the 128-wide gate gradients)
  int cell;
  const float* cin;     // c^r (rows of ld floats) or null (= zeros)
  float* dcdir;         // [N,32] in/out: gradient wrt the cell state
  int use_dcdir;
};
int spg_launch_ecc_step_bwd(const SpgEccStepBwd& p, hipStream_t stream);

// dW_e = sum_r h^r_src (x) g^r_dst  (matrix) / h^r_src * g^r_dst (vector), r = 0..R-1: one wavefront per edge, the sum over the
// iterations in registers, written once.  The body lives here so that the grouped-launch kernel (spg_gemm.hip) can run it as
// one of its jobs; bx = the workgroup's index inside the job.
struct SpgEdgeWgrad {
  SpgGraph g;
  int matrix, R;
  const float* states; long lds;
  const float* G; long ldg;
  float* dW;
};
#ifdef __HIPCC__
__device__ __forceinline__ void spg_ecc_edge_wgrad_body(const SpgEdgeWgrad& p, const int bx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = bx * 4 + wave;
  if (e >= p.g.E) return;
  const float* hs = p.states + (long)p.g.src[e] * p.lds;
  const float* gd = p.G + (long)p.g.dst[e] * p.ldg;
  if (p.matrix) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < p.R; ++r) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gd + r * 32 + 4 * (lane & 7));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float hk = hs[r * 32 + (lane >> 3) + 8 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(hk, g4[c], acc[q][c]);
      }
    }
    f32x4* o = reinterpret_cast<f32x4*>(p.dW + (long)e * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[lane + 64 * q] = acc[q];
  } else if (lane < 32) {
    float a = 0.f;
    for (int r = 0; r < p.R; ++r) a = fmaf(hs[r * 32 + lane], gd[r * 32 + lane], a);
    p.dW[(long)e * 32 + lane] = a;
  }
}
#endif
// spg_gemm.hip: true when the job was taken by the grouped launch that is open on this thread (nothing to launch then)
bool spg_group_add_edge_wgrad(const SpgEdgeWgrad& p, hipStream_t stream);
int spg_launch_ecc_edge_wgrad(const SpgGraph& g, int matrix, const float* states, long lds, const float* G, long ldg,
                              int R, float* dW, hipStream_t stream);
int spg_launch_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, hipStream_t stream);

// ---- persistent (one launch for all iterations) forms of the GRU recurrence, spg_ecc.hip ----
#define SPG_PX_WG_NODES 1024       // nodes per round with ONE 4-wave workgroup per CU (one wavefront per node, all co-resident)
#define SPG_PX_MAX_NODES 2048      // nodes per round with two workgroups per CU (fewer register-resident filters, gate rows from LDS)
#define SPG_PX_MAX_GROUPS 8        // rounds per launch: groups of whole connected components (scenes) of <= SPG_PX_MAX_NODES nodes
#define SPG_PX_MULTI_MAX_NODES 16000      // one group above SPG_PX_MAX_NODES: several nodes per wavefront, iteration-major (spg_ecc.hip)
inline bool spg_px_is_multi(int n_groups, int max_group) { return n_groups == 1 && max_group > SPG_PX_MAX_NODES; }
struct SpgPxGroups {               // node ranges [ptr[g], ptr[g+1]) of the rounds; n = 1: the whole graph in one round
  int n;
  int ptr[SPG_PX_MAX_GROUPS + 1];
};
// rounds for a batch whose connected components are unions of the node ranges part_ptr[0..n_parts] (scenes of a batch; n_parts
// = 0: unknown -- one round if the whole graph fits); false when a part exceeds a round or too many rounds would be needed
bool spg_px_plan_groups(int N, int n_parts, const int* part_ptr, SpgPxGroups* out);
#define SPG_PX_SAVE_F 12           // floats per lane and (node, iteration) of forward internals kept for the backward (3 quads)
// What follows the recurrence in the standard model -- the classifier Linear(nin -> C) on the module's output (h^R, or all of
// h^0 .. h^R with cat_all: nin = 32 or 32 (R + 1); learning/graphnet.py:47-49,81) and the weighted cross entropy of the trainer
// (learning/main.py:205) -- computed by the wavefront that owns the node: the logits grow by W[:, 32r : 32r + 32] h^r while the
// wave has h^r in LDS anyway (before it waits for its neighbours), log-sum-exp, the gradient wrt the logits and wrt the module's
// output follow its last iteration; the loss is summed by workgroup 0 of the NEXT launch (the persistent backward, at its end)
// with the summation order of ce_fwd_kernel (spg_loss.hip).  Replaces three latency-bound launches (classifier forward, cross entropy, classifier
// data gradient) between the two recurrences.  W == nullptr: no head.  The classifier's rows live in LDS (dynamic, C rows of
// nin + 4 floats): at most SPG_PX_HEAD_LDS bytes, so that two workgroups per CU stay resident.
#define SPG_PX_HEAD_MAXC 32
#define SPG_PX_HEAD_LDS 24576
struct SpgEccHead {
  const float* W;              // [C, nin]
  const float* b;              // [C] or null
  const int64_t* target;       // [N] class index or ignore_index
  const float* class_weight;   // [C] or null
  int64_t ignore_index;
  int C, reduction_mean, N, nin;
  float* logits;               // out [N, C]
  float* grad_logits;          // out [N, C]  d loss / d logits (for d loss = 1)
  float* grad_out;             // out [N, nin] d loss / d (module output)
  float* lse;                  // out [N] log-sum-exp per row
  float* loss;                 // out [1]
  float* wsum;                 // out [1] sum of the labelled rows' class weights
  float* dW;                   // out [C, nin] / [C] (or null): the classifier's parameter gradients, formed by SERVICE workgroups
  float* db;                   //   of the persistent backward launch when CUs are left for them (see node_wgs below)
};
inline size_t spg_px_head_lds_bytes(const SpgEccHead& h) { return h.W == nullptr ? 0 : (size_t)h.C * (size_t)(h.nin + 4) * sizeof(float); }
struct SpgEccPersistFwd {
  SpgGraph g;
  const float* W;
  int matrix, R;
  const float* h0;          // [N, 32]; with h0_rows: row i = h0[h0_rows[i]] (a zero row where < 0): the embedding scatter in place
  const int64_t* h0_rows;   // [N] or null
  float* states; long ldS;  // [N][(R+1)*32]: h^0 .. h^R (kept for the backward)
  float* agg;               // [N][(R+1)*32] aggregates per iteration, or null (inference)
  float* out; long ldo;     // cat_all: [N][(R+1)*32], else [N][32] = h^R
  int cat_all;
  SpgGruParams gru;
  unsigned long long* gran; // [SPG_PX_MAX_GROUPS][SPG_PX_MAX_ITERS][SPG_PX_MAX_NODES][32] granules
  unsigned* ctl;            // {epoch base, workgroups done, error count, -}
  SpgPxGroups groups;
  int gran_nodes;           // more nodes than wavefronts (one group above SPG_PX_MAX_NODES): nodes per iteration of the granule region (set by the launcher)
  float* fsave;             // [N][R][3 quads][64 lanes][4] forward internals kept for the backward (training), or null
  unsigned* fsave_tag;      // set to a magic word by the persistent forward when fsave was written
  SpgEccHead head;          // classifier + cross entropy behind the last iteration (head.W == nullptr: none)
  int node_wgs;             // = gridDim.x (set by the launcher)
};

struct SpgEccPersistBwd {
  SpgGraph g;
  const float* W;
  int matrix, R, cat_all;
  const float* grad_out; long ldgo;   // cat_all: [N][(R+1)*32]; else [N][32] (gradient wrt h^R)
  const float* states; long ldS;
  const float* agg;
  float* G;                 // [N][(R+1)*32]: gradient wrt the aggregate of iteration r (already / deg), slot r
  float* dgi; float* dgh; float* dui; float* duh; long ld96;    // [N][(R+1)*96]
  float* dpre; float* xg; long ld32;                            // [N][(R+1)*32]
  float* gx;                // [N, 32] gradient wrt h^0; with gx_rows: node j writes row gx_rows[j] (none where < 0)
  const int64_t* gx_rows;   // [N] or null
  SpgGruParams gru;
  unsigned long long* gran;
  unsigned* ctl;
  SpgPxGroups groups;
  int gran_nodes;           // (see SpgEccPersistFwd)
  const float* fsave;       // forward internals (see SpgEccPersistFwd) -- used when *fsave_tag carries the magic word
  const unsigned* fsave_tag;
  SpgEccHead head;          // the head the forward launch ran (W == nullptr: none)
  // workgroups [0, node_wgs) own nodes; workgroups beyond are SERVICE workgroups on CUs the recurrence leaves idle (set by the
  // launcher, within the residency bound; nobody waits for them): node_wgs sums the head's loss (without one, workgroup 0 does
  // at its end) and then, with all n_wgrad service workgroups, forms the classifier's weight / bias gradient: every wave takes
  // blocks of 64 columns (n_wgrad = 0: the caller queues them as a job of a grouped launch instead)
  int node_wgs, n_wgrad;
};

// return false when the persistent form is not applicable (too many nodes / iterations, switched off, another stream owns
// the exchange buffer): the caller then runs the per-iteration launches; *err != 0: the launch itself failed
bool spg_launch_ecc_persist_fwd(SpgEccPersistFwd p, hipStream_t stream, int* err);
// *head_wgrad_done (optional): the launch formed head.dW / head.db itself
bool spg_launch_ecc_persist_bwd(SpgEccPersistBwd p, hipStream_t stream, int* err, bool* head_wgrad_done = nullptr);

// spg_train_step: CloudEmbedder's scatter of the embeddings to all superpoints (learning/pointnet.py:177-179) and the gather of
// their gradients, fused into the one-launch recurrence (it reads row slot_of_row[i] of `emb`, a zero row where < 0, and writes
// the gradient of node j to row slot_of_row[j] of grad_emb); the per-iteration fallback materialises desc / gathers grad_desc
struct SpgEccScatter {
  const float* emb;               // [B, 32]
  const int64_t* slot_of_row;     // [N]
  const int64_t* idx_valid;       // [B]
  float* desc;                    // [N, 32] (fallback only)
  float* grad_emb;                // [B, 32]
  int B;
};

// device address of {error word, withheld-update counter} of the current device's persistent-launch control block, or null while
// no persistent launch has run on it (spg_ecc.hip); read by the guarded optimiser step (spg_api.hip)
unsigned* spg_px_guard_words();
