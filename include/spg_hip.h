/*
 * spg_hip.h -- C ABI of libspg_hip.so, the MI355X (gfx950 / CDNA4) implementation of the
 * superpoint-graph learning hot path of loicland/superpoint_graph:
 *   PointNet superpoint embedding + edge-conditioned graph convolution (ECC) with GRU update,
 *   forward and backward.
 *
 * The reference has no FFI registry; its boundary to native code is one call site
 * (learning/ecc/cuda_kernels.py:117-139, invoked from learning/ecc/GraphConvModule.py:78-80,110-112)
 * plus stock torch.nn calls.  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer borrowed for the duration of
 *     the call; the caller -- torch -- owns all memory and sizes workspaces with the *_workspace_bytes queries.  The
 *     library itself owns exactly one device allocation per GPU: 16 KB of self-resetting reduction tickets (first launch of a
 *     sliced BatchNorm finalize), plus -- only after spg_rccl_init -- what RCCL allocates for its communicator;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return value: 0 on success, a hipError_t (>0) if a launch failed, -1 for an argument error;
 *     spg_last_error() returns the message (thread local);
 *   - matrices are row-major fp32 unless stated; index buffers are int64 exactly as
 *     learning/ecc/GraphConvInfo.py:62-69 produces them.
 */
#ifndef SPG_HIP_H
#define SPG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPG_MAX_LAYERS 8
#define SPG_VERSION 200

const char* spg_last_error(void);
int spg_version(void);

/* ------------------------------------------------------------------------------------------------
 * Graph structure.  Replaces GraphConvInfo.cuda() (learning/ecc/GraphConvInfo.py:71-76) and the
 * per-call torch.cumsum(degs) of learning/ecc/cuda_kernels.py:123,135: builds, on the device, the CSR
 * by target (rowptr, dst) and the reverse CSR by source used by the atomic-free backward.
 * idxn: int64 [E] source node per edge (edges sorted by target); degs: int64 [N] in-degrees.
 * n_src >= N: number of rows of the INPUT feature matrix (GraphConvFunction allows more input rows than
 * output nodes; for superpoint graphs n_src == N).  Indices outside [0, n_src) raise the header's error flag.
 * ---------------------------------------------------------------------------------------------- */
size_t spg_graph_workspace_bytes(int N, int n_src, int E);
int spg_graph_build(const int64_t* idxn, const int64_t* degs, int N, int n_src, int E, void* graph_ws, void* stream);
/* debug / test view: copies of the derived int32 arrays (device pointers, sizes N+1, E, E, n_src+1, E) and of the
 * header {N, n_src, E, error flag} */
int spg_graph_export(const void* graph_ws, int N, int n_src, int E, int32_t* rowptr, int32_t* src, int32_t* dst,
                     int32_t* rev_rowptr, int32_t* rev_eid, int32_t* hdr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generic ECC operator = GraphConvFunction.forward/backward (learning/ecc/GraphConvModule.py:44-152)
 * including the segment mean of conv_aggregate_fw/bw (learning/ecc/cuda_kernels.py:55-139).
 * dtype: 0 = float32, 1 = float64.  x [n_x_rows, cin]; w [n_w_rows, cin, cout] (w_is_matrix) or
 * [n_w_rows, cin] (cin == cout); idxe: int64 [E] filter index per edge or NULL (identity);
 * out / grad_out [N, cout]; grad_x [n_x_rows, cin]; grad_w like w.  Results do not depend on the
 * reference's edge_mem_limit sharding, so there is no such argument.
 * ---------------------------------------------------------------------------------------------- */
int spg_ecc_aggregate_fwd(int dtype, const void* x, const void* w, const int64_t* idxe, const void* graph_ws, int N,
                          int E, int cin, int cout, int w_is_matrix, void* out, void* stream);
int spg_ecc_aggregate_bwd(int dtype, const void* x, const void* w, const int64_t* idxe, const void* graph_ws, int N,
                          int E, int n_x_rows, int n_w_rows, int cin, int cout, int w_is_matrix, const void* grad_out,
                          void* grad_x, void* grad_w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRUCellEx.forward / backward (learning/modules.py:224-251), hidden = input = 32 channels.
 * params: 6 pointers {weight_ih [96,32], weight_hh [96,32], bias_ih [96], bias_hh [96],
 * ig.weight [32,32], ig.bias [32]} (the last two may be NULL when ingate == 0).
 * scratch: >= spg_gru_scratch_floats(n) floats.  grads: 6 pointers like params.
 * ---------------------------------------------------------------------------------------------- */
size_t spg_gru_scratch_floats(int n_rows);
int spg_gru_cell_fwd(const float* input, const float* hidden, int n_rows, const float* const* params, int layernorm,
                     int ingate, float* out, float* scratch, void* stream);
int spg_gru_cell_bwd(const float* input, const float* hidden, const float* grad_out, int n_rows,
                     const float* const* params, int layernorm, int ingate, float* grad_input, float* grad_hidden,
                     float* const* grads, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LSTMCellEx.forward / backward (learning/modules.py:280-309), hidden = input = 32 channels.
 * params: 6 pointers {weight_ih [128,32], weight_hh [128,32], bias_ih [128], bias_hh [128], ig.weight [32,32],
 * ig.bias [32]}.  hidden = (h, c); outputs hy, cy.  Backward takes the gradients wrt hy and cy (either may be NULL =
 * zero) and returns the gradients wrt input, h and c.  scratch: >= spg_lstm_scratch_floats(n) floats.
 * ---------------------------------------------------------------------------------------------- */
size_t spg_lstm_scratch_floats(int n_rows);
int spg_lstm_cell_fwd(const float* input, const float* h, const float* c, int n_rows, const float* const* params,
                      int layernorm, int ingate, float* hy, float* cy, float* scratch, void* stream);
int spg_lstm_cell_bwd(const float* input, const float* h, const float* c, const float* grad_hy, const float* grad_cy,
                      int n_rows, const float* const* params, int layernorm, int ingate, float* grad_input,
                      float* grad_h, float* grad_c, float* const* grads, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused dense layer on the MFMA row-GEMM core (replaces nn.Linear / nn.Conv1d(k=1) + the preceding
 * BatchNorm/ReLU, e.g. learning/graphnet.py:47-49):  Y = act(X * in_scale + in_shift) @ W^T + bias.
 * X [M, ldx] (first K columns used), W [N, K], in_scale / in_shift [K] or NULL, Y [M, ldy].
 * spg_linear_wgrad: dW [N, K] = dY^T @ act(X*in_scale+in_shift); work >= spg_linear_wgrad_work_floats.
 * ---------------------------------------------------------------------------------------------- */
int spg_linear_fwd(const float* X, long ldx, int M, int K, const float* W, const float* bias, int N,
                   const float* in_scale, const float* in_shift, int in_relu, float* Y, long ldy, void* stream);
/* dX[M,K] = dY[M,N] @ W[N,K] (data gradient of the layer above) and out[N] = column sums of X[M,N] (bias gradient;
 * work >= 64*N floats): the remaining pieces of nn.Linear's backward (learning/graphnet.py:47-49, the classifier). */
int spg_linear_dgrad(const float* dY, long lddy, int M, int N, const float* W, int K, float* dX, long ldx, void* stream);
int spg_colsum(const float* X, long ldx, long M, int N, float* out, float* work, void* stream);
size_t spg_linear_wgrad_work_floats(int M, int N, int K);
int spg_linear_wgrad(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K, const float* in_scale,
                     const float* in_shift, int in_relu, float* dW, float* work, void* stream);
/* weight and bias gradient of a dense layer together (the column sums of dY ride along with the weight-gradient launch):
 * dbias [N] = sum_m dY[m, :]; work >= spg_linear_wgrad_bias_work_floats(M, N, K) floats.  Replaces the autograd backward
 * of nn.Linear's weight and bias (classifier, learning/graphnet.py:47-49). */
size_t spg_linear_wgrad_bias_work_floats(int M, int N, int K);
int spg_linear_wgrad_bias(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K, const float* in_scale,
                          const float* in_shift, int in_relu, float* dW, float* dbias, float* work, void* stream);
/* The whole backward of Y = X W^T + b in ONE grouped launch + one batched reduction (round 4): dX [M, lddx] (may be null),
 * dW [N, K], dbias [N] (may be null) -- the three are mutually independent, so they run as jobs of a single kernel
 * (autograd of nn.Linear, the classifier of learning/graphnet.py:47-49).  work as for spg_linear_wgrad_bias. */
int spg_linear_backward(const float* dY, long lddy, const float* X, long ldx, const float* W, int M, int N, int K, float* dX,
                        long lddx, float* dW, float* dbias, float* work, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PointNet (learning/pointnet.py:16-133): STNkD + per-point MLP + max-pool + FC head, train-mode
 * BatchNorm statistics over the whole batch, eval mode from the running statistics.
 * Replaces PointNet.forward / its autograd backward (and the recompute of CloudEmbedder.bw_hook,
 * learning/pointnet.py:160-176: activations are kept instead, the running statistics are updated
 * `bn_update_times` times to reproduce the double forward).
 *
 * params / grads: one group of 6 pointers per layer, in this layer order:
 *   stn.convs[0..n_stn_conv), stn.fcs[0..n_stn_fc), stn.proj, convs[0..n_conv), fcs[0..n_fc)
 * group = {weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var}; absent entries NULL
 * (grads: {d weight, d bias, d bn.weight, d bn.bias, NULL, NULL}).  With nfeat_stn == 0 the STN
 * groups are omitted.
 * clouds [B, nfeat, npts] fp32, clouds_global [B, nfeat_global]; emb [B, fc[n_fc-1]].
 * ---------------------------------------------------------------------------------------------- */
typedef struct spg_pointnet_cfg {
  int nfeat, nfeat_stn, nfeat_global, npts;
  int n_stn_conv, n_stn_fc, n_conv, n_fc;
  int stn_conv[SPG_MAX_LAYERS], stn_fc[SPG_MAX_LAYERS], conv[SPG_MAX_LAYERS], fc[SPG_MAX_LAYERS];
  int last_ac;          /* BatchNorm+ReLU after the last FC (PointNet(last_ac=True)) */
  float bn_eps, bn_momentum;
} spg_pointnet_cfg;

int spg_pointnet_num_layers(const spg_pointnet_cfg* cfg);
size_t spg_pointnet_workspace_bytes(const spg_pointnet_cfg* cfg, int B, int training);
int spg_pointnet_forward(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                         const void* const* params, float* emb, void* workspace, int training, int bn_update_times,
                         void* stream);
/* test helper: byte offset of a layer's buffer inside the forward workspace (what: 0 raw output, 1 BN scale,
 * 2 BN shift, 3 batch mean, 4 batch rstd; layer -1 / -2 = STN / main segment: what 0 pooled raw values [B, ld],
 * 1 arg-max points int32 [B, ld], 2 the leading dimension ld itself); -1 if absent */
long spg_pointnet_debug_offset(const spg_pointnet_cfg* cfg, int B, int training, int layer, int what);
size_t spg_pointnet_bwd_workspace_bytes(const spg_pointnet_cfg* cfg, int B);
int spg_pointnet_backward(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                          const void* const* params, const float* grad_emb, void* const* grads, void* workspace,
                          void* bwd_workspace, void* stream);
/* The same with an EXTERNALLY evaluated spatial transformer, as LocalCloudEmbedder.run_batch uses the networks
 * (learning/pointnet.py:182-205: a stand-alone STNkD, then a PointNet built with nfeat_stn = 0).  ext_transform: [B, 4] =
 * T - I (row-major 2x2 per cloud); the first convolution applies [x y] @ T while it stages the cloud, so the
 * transformed clouds never exist in HBM.  Backward: grad_transform [B, 4] (gradient wrt T) and grad_global
 * [B, nfeat_global] (gradient wrt clouds_global, into which the reference concatenates T) are optional outputs. */
int spg_pointnet_forward_ext(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                             const float* ext_transform, const void* const* params, float* emb, void* workspace, int training,
                             int bn_update_times, void* stream);
int spg_pointnet_backward_ext(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                              const float* ext_transform, const void* const* params, const float* grad_emb, void* const* grads,
                              float* grad_transform, float* grad_global, void* workspace, void* bwd_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RNN-ECC module = RNNGraphConvModule.forward (learning/modules.py:152-183) with a GRUCellEx or LSTMCellEx cell:
 * filter-generating MLP on the edge features (learning/graphnet.py:17-34, once per forward), then
 * nrepeats x { ECC aggregate (GraphConvFunction) ; GRU }, states concatenated when cat_all.
 * params / grads groups (6 pointers each, as above): fnet linear layers [0..n_fnet) (the BatchNorm
 * after layer `bnidx` lives in that layer's group), then one group for the cell:
 *   {weight_ih, weight_hh, bias_ih, bias_hh, ig.weight, ig.bias}.
 * h0 [N, nc] (nc must be 32), edgefeats [E, fnet_widths[0]]; out [N, nc*(nrepeats+1)] (cat_all) or [N, nc].
 * ---------------------------------------------------------------------------------------------- */
#define SPG_MAX_PARTS 64
typedef struct spg_eccrnn_cfg {
  int nc, nrepeats, matrix, layernorm, ingate, cat_all;
  int n_fnet;                           /* number of Linear layers of the filter network */
  int fnet_widths[SPG_MAX_LAYERS + 1];  /* n_fnet + 1 entries: input width ... output width */
  int bnidx, llbias;
  float bn_eps, bn_momentum;
  int cell;                             /* 0: GRUCellEx (gru_*), 1: LSTMCellEx (lstm_*; weights [128,32], cx starts at 0,
                                           learning/modules.py:168-169) */
  /* Per-batch hint (0 = unknown): the batched graph's nodes [part_ptr[k], part_ptr[k+1]) , k < n_parts, are closed under
   * edges -- the scenes of a batch (learning/spg.py:178-193 concatenates them with node offsets).  With it the
   * dataflow-synchronised GRU recurrence (all iterations in ONE launch) also serves batches of more than 2048 nodes: whole
   * scenes are packed into rounds of <= 2048 nodes.  Without it (or when a scene alone exceeds a round, or more than 8 rounds
   * are needed) graphs of up to 16 000 nodes run as ONE group with several nodes per wavefront, iteration-major (round 5:
   * Semantic3D-scale scenes are one component of ~10 000 superpoints); beyond that, one launch per iteration.  Results are
   * identical in all forms.  Must be the same in the workspace-size queries, the
   * forward and the backward of a batch. */
  int n_parts;
  int part_ptr[SPG_MAX_PARTS + 1];
} spg_eccrnn_cfg;

size_t spg_eccrnn_workspace_bytes(const spg_eccrnn_cfg* cfg, int N, int E, int training);
int spg_eccrnn_forward(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* h0,
                       const float* edgefeats, const void* const* params, float* out, void* workspace, int training,
                       int bn_update_times, void* stream);
/* test helper: byte offset inside the forward workspace of filter-network layer `layer`'s buffers (what: 0 raw output
 * [E, cout], 1 / 2 BatchNorm scale / shift of that layer); -1 if absent */
long spg_eccrnn_debug_offset(const spg_eccrnn_cfg* cfg, int N, int E, int training, int layer, int what);
size_t spg_eccrnn_bwd_workspace_bytes(const spg_eccrnn_cfg* cfg, int N, int E);
int spg_eccrnn_backward(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* edgefeats,
                        const void* const* params, const float* grad_out, float* grad_h0, void* const* grads,
                        void* workspace, void* bwd_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side superpoint loader: the body of load_superpoint (learning/spg.py:198-236: resample to npts rows,
 * centre / normalise xyz, select the feature columns, transpose) and the geometric part of augment_cloud
 * (:239-258) for ALL superpoints of a ragged point buffer in one launch.
 *   points [Ntot, ncols] raw rows (xyz first), offsets i64 [S+1] first row of each superpoint,
 *   slot i32 [S]: output row of the superpoint, or -1 when it has fewer than ptn_minpts points (spg.py:203),
 *   sample_idx i32 [S, npts]: row indices inside the superpoint = the reference's `rs.choice` stream (:207-214),
 *   colmap i32 [nfeat]: raw column of each output feature (the pc_attribs selection, :224-232; 0..2 = normalised xyz),
 *   M f64 [S, 3, 3] or NULL: augmentation matrix, applied as P[:, :3] @ M^T (:252), noise [Nv, npts, nfeat] or NULL: jitter,
 *   clouds [Nv, nfeat, npts] (the layout spg_pointnet_forward consumes), diam [Nv] (0 when xyznormalize == 0).
 * ---------------------------------------------------------------------------------------------- */
#define SPG_LOADER_MAX_FEATS 16
int spg_load_superpoints(const float* points, int ncols, const int64_t* offsets, int n_superpoints, const int32_t* slot,
                         const int32_t* sample_idx, int npts, int xyznormalize, const int32_t* colmap, int nfeat,
                         const double* M, const float* noise, float* clouds, float* diam, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batch construction on the device.
 * spg_set_batch = GraphConvInfo.set_batch (learning/ecc/GraphConvInfo.py:33-69) for edges already carrying their batch
 * node offsets: edges int64 [E][2] (source, target) -> idxn int64 [E] (source per edge, edges ordered by target),
 * degs int64 [N] (in-degree), perm int64 [E] (original index of the edge at every position: apply it to per-edge
 * data with spg_gather_rows, GraphConvInfo.py:53-56).  The order is the STABLE sort by target (deterministic); the
 * reference uses numpy's default argsort, whose tie order is unspecified.  error_flag (device int32, may be NULL)
 * becomes 1 when an endpoint is outside [0, N).  workspace: >= spg_set_batch_workspace_bytes(N, E) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t spg_set_batch_workspace_bytes(int N, int E);
int spg_set_batch(const int64_t* edges, int N, int E, int64_t* idxn, int64_t* degs, int64_t* perm, void* workspace,
                  int32_t* error_flag, void* stream);
/* dst[r, :cols] = src[perm[r], :cols]; a negative perm[r] gives a zero row (CloudEmbedder's scatter of the embeddings of
 * the valid superpoints into the zero-initialised descriptor matrix, learning/pointnet.py:177-179, as ONE launch) */
int spg_gather_rows(const float* src, long ld_src, const int64_t* perm, long rows, int cols, float* dst, long ld_dst,
                    void* stream);

/* spg_edge_features + scaler01's transform (learning/spg.py:23-64).  One spec per OUTPUT column: COPY takes column
 * `column` of a per-edge float32 matrix (delta_avg / delta_std); DIFF / LOGDIFF / RATIO combine the per-node attribute
 * `data[node, column]` of the edge's source and target as attr[s]-attr[t] / log(attr[s]+1e-10)-log(attr[t]+1e-10) /
 * attr[s]/(attr[t]+1e-10); CONST = 1.  is_f64: the attribute matrix is float64 (the u64 point count promoted like
 * numpy does) and the arithmetic is float64 with one final rounding; otherwise float32 arithmetic.  mean / scale
 * (float64 [ncols], device, or both NULL): sklearn StandardScaler.transform on the float32 result. */
enum { SPG_EF_COPY = 0, SPG_EF_DIFF = 1, SPG_EF_LOGDIFF = 2, SPG_EF_RATIO = 3, SPG_EF_CONST = 4 };
#define SPG_EF_MAX_COLS 32
typedef struct {
  const void* data;
  long ld;
  int column, kind, is_f64, pad_;
} spg_edge_feature_spec;
typedef struct {
  int ncols, pad_;
  spg_edge_feature_spec col[SPG_EF_MAX_COLS];
} spg_edge_feature_specs;
int spg_edge_features(const spg_edge_feature_specs* specs, const int64_t* edges, long E, const double* mean,
                      const double* scale, float* out, void* stream);

/* The whole construction of a SMALL batch (N <= 4096 nodes, E <= 32768 edges) in two launches: spg_set_batch +
 * spg_gather_rows of the edge features + spg_graph_build, from HOST pointers (edges int64 [E][2] with the batch node offsets
 * applied, feats float32 [E][F]; uploaded through the staging ring of spg_upload) -> idxn [E], degs [N], feats_sorted [E][F]
 * (GraphConvInfo.get_buffers(), learning/ecc/GraphConvInfo.py:33-79) and the device graph in graph_ws
 * (spg_graph_workspace_bytes(N, N, E)).  Same results as the three calls it replaces (stable order by target).
 * spg_batch_graph_scratch_bytes returns 0 when the batch is too large for it: use the three calls then. */
size_t spg_batch_graph_scratch_bytes(int N, int E, int F);
int spg_batch_graph_build(const int64_t* edges_host, const float* feats_host, int N, int E, int F, int64_t* idxn, int64_t* degs,
                          float* feats_sorted, void* graph_ws, void* scratch, int32_t* error_flag, void* stream);
/* (round 6) The same with the edge list and the edge features ALREADY on the device -- uploaded together with the batch's other
 * small vectors by one spg_upload_packed: a fresh batch then costs the host two copies (the clouds, everything else) instead of
 * seven (learning/main.py:189-203, learning/spg.py:178-193). */
int spg_batch_graph_build_dev(const int64_t* edges_dev, const float* feats_dev, int N, int E, int F, int64_t* idxn, int64_t* degs,
                              float* feats_sorted, void* graph_ws, void* scratch, int32_t* error_flag, void* stream);
/* n host buffers -> one page-locked staging slot -> ONE asynchronous host-to-device copy: piece i lands at (char*)device + offsets[i]
 * (the caller's layout: non-overlapping, inside `total` bytes).  Returns once the copy is enqueued; the host buffers may be re-used. */
int spg_upload_packed(const void* const* host, const size_t* bytes, const size_t* offsets, int n, void* device, size_t total, void* stream);

/* Host -> device upload of a small per-batch buffer (the reference's `.cuda()` of idxn / degs / edgefeats,
 * learning/ecc/GraphConvInfo.py:71-79, and of the clouds, learning/pointnet.py:150-152) WITHOUT a host stall: the bytes
 * are copied into a slot of a library-owned ring of page-locked staging buffers and sent from there with an asynchronous
 * copy on `stream`; the call returns as soon as the copy is enqueued and `host` may be re-used at once.  A copy from
 * pageable memory would block the host until the stream reaches it (= until the previous step has drained).  A slot is
 * re-used only after the event recorded behind its last copy has completed.  Thread-safe.  bytes == 0 is a no-op. */
int spg_upload(const void* host, size_t bytes, void* device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Superpoint-graph construction on the device (SURVEY.md section 8, row f4 tail): partition/graphs.py:75-210
 * compute_sp_graph AFTER scipy's Delaunay triangulation (which stays on the host), and compute_geof of
 * partition/ply_c/ply_c.cpp:384-462.  Host mirror: superpoint_graph_amd/partition/graphs.py.
 *   spg_spg_tet_edges      tets int32 [T,4] (Delaunay.simplices), comp int32 [n] (in_component) -> both directions of every
 *                          vertex pair that joins two components (graphs.py:85-107) as keys (source << 32 | target), in
 *                          arbitrary order; count (device uint64) = number of keys found (keys may be NULL: count only);
 *   spg_spg_unique_edges   np.unique(edges, axis=1) + the d_max filter (:108-113; float32 distance in the reference's
 *                          operation order, d_max <= 0: no filter) -> edge_keys [count] in (source, target) order and their
 *                          component-pair index comp[s] * n_com + comp[t] (:120);
 *   spg_spg_group_edges    stable order by component pair (:121-125) and the superedge segments (:127-128): seg_cc [n_seg],
 *                          seg_off [n_seg + 1] (capacity n + 1), n_seg (device int64);
 *   spg_spg_superpoints    :141-172 for all components: centroids [n_com,3], length / surface / volume [n_com], point_count
 *                          uint64 [n_com], sp_labels uint32 [n_com, n_labels + 1] from labels int32 [n] (histogram) or
 *                          label_rows uint32 [n, n_labels + 1] (column sums), or neither;
 *   spg_spg_superedges     :174-208 for all superedges.
 * workspace: spg_spg_workspace_bytes(which = 0 unique_edges / 1 group_edges / 2 superpoints, n = number of keys / edges / points).
 * Integer results are bit-exact; float features are accumulated in float64 and rounded once.
 * ---------------------------------------------------------------------------------------------- */
size_t spg_spg_workspace_bytes(int which, long n);
int spg_spg_tet_edges(const int32_t* tets, long T, const int32_t* comp, uint64_t* keys, long capacity, uint64_t* count, void* stream);
int spg_spg_unique_edges(const uint64_t* keys, long n, const float* xyz, const int32_t* comp, long n_com, float d_max,
                         uint64_t* edge_keys, uint64_t* cc_keys, uint64_t* count, void* workspace, size_t workspace_bytes, void* stream);
int spg_spg_group_edges(const uint64_t* cc_keys, const uint64_t* edge_keys, long n, uint64_t* cc_sorted, uint64_t* edges_sorted,
                        uint64_t* seg_cc, int64_t* seg_off, int64_t* n_seg, void* workspace, size_t workspace_bytes, void* stream);
int spg_spg_superpoints(const float* xyz, long n, const int32_t* comp, int n_com, const int32_t* labels, const uint32_t* label_rows,
                        int n_labels, float* centroids, float* length, float* surface, float* volume, uint64_t* point_count,
                        uint32_t* sp_labels, void* workspace, size_t workspace_bytes, void* stream);
int spg_spg_superedges(const uint64_t* edges_sorted, const uint64_t* seg_cc, const int64_t* seg_off, long n_sedg, long n_com,
                       const float* xyz, const float* centroids, const float* length, const float* surface, const float* volume,
                       const uint64_t* point_count, uint32_t* source, uint32_t* target, float* delta_mean, float* delta_std,
                       float* delta_norm, float* delta_centroid, float* length_ratio, float* surface_ratio, float* volume_ratio,
                       float* point_count_ratio, void* stream);
/* compute_geof (ply_c.cpp:384-462): xyz float32 [n,3], target uint32 [n, k_nn] (the k_nn nearest neighbours of every point)
 * -> geof float32 [n,4] = linearity, planarity, scattering, verticality.  Covariance and eigen-decomposition in float64. */
int spg_compute_geof(const float* xyz, const uint32_t* target, long n, int k_nn, float* geof, void* stream);

/* prune (partition/ply_c/ply_c.cpp:288-382, libply_c.prune(xyz, voxel_size, rgb, labels, objects, n_labels, n_objects)):
 * regular voxel grid -> per non-empty voxel the mean position, the mean colour and the label / object histograms; voxels are
 * numbered in the order of their first point.  Two phases because the number of voxels sizes the outputs: spg_prune_voxels
 * (bins, ordering; *n_voxels on the device) then spg_prune_reduce on the SAME workspace.  error flag: 1 = more than 2^21 bins
 * along an axis, 2 = a label / object id beyond n_labels / n_objects.  Float results use the reference's float32 operations in
 * the reference's order (bit-exact restatement; the reference itself cannot be built here: Boost). */
size_t spg_prune_workspace_bytes(long n);
int spg_prune_voxels(const float* xyz, long n, float voxel_size, int64_t* n_voxels, int32_t* error_flag, void* workspace,
                     size_t workspace_bytes, void* stream);
int spg_prune_reduce(const float* xyz, const uint8_t* rgb, const uint8_t* labels, const uint32_t* objects, long n, long n_voxels,
                     int n_labels, int n_objects, float* out_xyz, uint8_t* out_rgb, uint32_t* out_labels, uint32_t* out_objects,
                     int32_t* error_flag, void* workspace, size_t workspace_bytes, void* stream);

/* Random streams of the loader generated on the device (optional; the default keeps numpy's streams on the host so
 * that seeded runs reproduce the reference's clouds): Philox4x32-10 keyed by (seed, superpoint id, step).  counts /
 * ids int64 [S], slot int32 [S] (row of the cloud tensor or -1) -> sample_idx int32 [S, npts] (spg.py:207-214), M
 * float64 [S, 3, 3] (augment_cloud's matrix, :241-251; may be NULL) and noise float32 [n_valid, npts, nfeat] (clipped
 * N(0, 0.01^2), :255-257; may be NULL), the inputs of spg_load_superpoints. */
int spg_loader_random(const int64_t* counts, const int64_t* ids, const int32_t* slot, int n_superpoints, int npts, int nfeat,
                      uint64_t seed, uint32_t step, int augment, float scale, int rot, float mirror_prob, int jitter,
                      int32_t* sample_idx, double* M, float* noise, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weighted cross entropy of the training / evaluation loops (learning/main.py:205,255:
 * nn.functional.cross_entropy(outputs, label_mode, weight=class_weights); rows with target == ignore_index do not
 * count): loss (1 float; reduction_mean: sum_i w[t_i] nll_i / sum_i w[t_i], else the plain sum), lse [N] (log-sum-exp
 * per row, kept for the backward), wsum (1 float, the normaliser).  Backward: grad_logits [N, C] from grad_loss (1 float).
 * One launch each, fixed summation order.  A target outside [0, C) that is not ignore_index (torch: device-side assert)
 * makes the loss NaN -- the kernels are asynchronous, so the error surfaces in the value instead of an error code.
 * ---------------------------------------------------------------------------------------------- */
int spg_cross_entropy_fwd(const float* logits, const int64_t* target, const float* weight, int N, int C, int64_t ignore_index,
                          int reduction_mean, float* loss, float* lse, float* wsum, void* stream);
int spg_cross_entropy_bwd(const float* logits, const int64_t* target, const float* weight, const float* lse, const float* wsum,
                          const float* grad_loss, int N, int C, int64_t ignore_index, int reduction_mean, float* grad_logits,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation accounting on the device (learning/main.py:246-263, eval_final :267-311, metrics.py:16-18):
 * logits [n_samples][N][C] (sample_stride floats between samples; the mean over the test-time samples is taken in
 * float32 in sample order like np.mean), pred i64 [N] = first arg-max per superpoint, and for superpoints with
 * label_mode != -100: confusion i64 [C][C] (ground truth x predicted) += label_vec[i, :] in column pred_i,
 * counters[0] += (pred == label_mode), counters[1] += 1.  Exact integer atomics; confusion / counters accumulate.
 * ---------------------------------------------------------------------------------------------- */
int spg_eval_accumulate(const float* logits, int n_samples, long sample_stride, int N, int C, const int64_t* label_mode,
                        const int64_t* label_vec, int64_t* pred, int64_t* confusion, int64_t* counters, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Synchronised BatchNorm for the data-parallel mode (the reference's single process normalises over
 * ALL scenes of the batch, learning/pointnet.py:34-47,93-108, graphnet.py:26-27).  The library owns no communicator:
 * the host registers an all-reduce(SUM) callback over a caller-owned fp64 DEVICE buffer.  While registered, every
 * train-mode BatchNorm of spg_pointnet_* / spg_eccrnn_* (forward statistics and the two backward sums) runs as
 *   local reduction into buf -> fn(ctx, buf, n, stream) -> finish from the global sums,
 * n <= 3*C+1 doubles for a C-channel layer.  fn must enqueue the collective so that it is ordered after the work
 * already on `stream` and before later work on it (torch.distributed semantics), and return 0.  All ranks must
 * run the same layer sequence.  fn == NULL restores per-rank statistics (the default).
 * ---------------------------------------------------------------------------------------------- */
typedef int (*spg_allreduce_fn)(void* ctx, double* buf, long n, void* stream);
int spg_set_bn_allreduce(spg_allreduce_fn fn, void* ctx, double* buf, long buf_doubles);
/* Slot-synchronised BatchNorm (round 5; the same semantics -- statistics over the scenes of ALL ranks -- at the speed of the
 * per-rank mode): train-mode BatchNorm statistics travel from producer to consumer launch as exact 64-bit fixed-point sums
 * ("slots"); while a slot all-reduce is registered, the library calls
 *   fn(ctx, words, nwords, stream)      -- element-wise SUM of `nwords` int64 words over the ranks, in place, in stream order
 * behind every launch that produced such sums and before the launch that consumes them.  Integer sums are exact and
 * order-independent: every rank holds bit-identical statistics, whatever the rank count.  Everything built on the slots keeps
 * running (statistics folds, fused convolution backward, one-pass first layers, grouped launches, spg_train_step).
 * The row counts the consumers divide by travel with the sums (a spare word of every slot array counts the rows of its producer
 * launches), so the all-reduce delivers the rows of all ranks too.  world: number of ranks (capacity check of the slots; the
 * BatchNorm parameter gradients, formed by every rank from the global sums, are divided by it).  fn == NULL switches the mode off.
 * All ranks must run the same launch sequence; a rank without edges cannot take part.  Mutually exclusive with
 * spg_set_bn_allreduce. */
typedef int (*spg_slot_allreduce_fn)(void* ctx, unsigned long long* words, long nwords, void* stream);
int spg_set_slot_allreduce(spg_slot_allreduce_fn fn, void* ctx, int world);

/* RCCL from inside the library: a communicator of the library's own, bootstrapped by the host --
 * rank 0 obtains 128 bytes with spg_rccl_unique_id and distributes them (any channel), every rank calls spg_rccl_init
 * with its HIP device current.  spg_rccl_allreduce_sum_f32: in-place sum all-reduce on `stream` (the flat gradient
 * arena: ONE collective per step).  spg_rccl_sync_bn(buf, n): the synchronised-BatchNorm all-reduces go through this
 * communicator, enqueued by the library between the two halves of every BatchNorm finalize (buf as in
 * spg_set_bn_allreduce; NULL switches the mode off).  librccl is bound with dlopen at the first call. */
int spg_rccl_unique_id(void* out_128_bytes);
int spg_rccl_init(const void* unique_id_128_bytes, int world_size, int rank);
int spg_rccl_world_size(void);
int spg_rccl_allreduce_sum_f32(float* buf, long n, void* stream);
int spg_rccl_sync_bn(double* buf, long buf_doubles);
/* slot-synchronised BatchNorm (spg_set_slot_allreduce) through the library's communicator: the slots are summed in place as
 * 64-bit integers (ncclInt64); on = 0 switches the mode off */
int spg_rccl_sync_slots(int on);
int spg_rccl_allreduce_sum_f64(double* buf, long n, void* stream);
int spg_rccl_destroy(void);

/* ------------------------------------------------------------------------------------------------
 * Element-wise gradient clamp + Adam step on one flat parameter buffer: replaces the per-parameter loop
 * `p.grad.data.clamp_(-clip, clip)` (learning/main.py:210-212) and `optimizer.step()` of torch.optim.Adam
 * (learning/main.py:213, :433-437) by ONE launch.  `step` counts from 1.  grad_clip <= 0 disables the clamp.
 * FAIL-SAFE (round 6): the launch is a no-op while the time-out word of the device's one-launch RNN-ECC recurrences is non-zero
 * (spg_ecc_persistent_status) -- gradients computed from stale neighbour states never reach the parameters.
 * ---------------------------------------------------------------------------------------------- */
int spg_adam_clamp_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float grad_clip, int step, void* stream);
/* the same with every gradient divided by *grad_div (one DEVICE float, may be NULL) before the clamp: the data-parallel
 * normaliser -- the all-reduced sum of the ranks' loss weights -- is consumed where the collective left it, without a host
 * round trip or a separate scaling launch */
int spg_adam_clamp_step_scaled(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float grad_clip, int step, const float* grad_div,
                               void* stream);
/* (round 6) The same with a second, caller-supplied guard: *guard_all (a device float, may be null) != 0 withholds the update like the
 * device's own time-out word does.  Data parallel: spg_ecc_persistent_flag writes this rank's flag (0 / 1) next to the gradients, the
 * gradient all-reduce sums the flags, every rank passes the sum here -- one rank's time-out withholds the update on EVERY rank (the
 * summed gradients contain its wrong ones), so the replicas stay identical (superpoint_graph_amd/flat.py: allreduce_sums). */
int spg_adam_clamp_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float grad_clip, int step, const float* grad_div,
                                const float* guard_all, void* stream);
int spg_ecc_persistent_flag(float* dst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Instrumentation for bench.py: when enabled, every launch of the row-GEMM kernels is bracketed by
 * hipEvents on its own stream; spg_prof_read synchronises and returns the accumulated milliseconds,
 * launch count and algorithmic FLOPs since the last reset.
 * ---------------------------------------------------------------------------------------------- */
void spg_prof_enable(int on);
int spg_prof_read(double* ms, long* launches, double* flops, int reset);
/* the same, restricted to one kernel instantiation: tag = spg_prof_tag(kind, IT, JT, x, y, full) with kind 1 = row-GEMM
 * (x = 1 for the data-gradient form, y = operand mode of A) or 2 = weight gradient (x, y = operand modes of A and B);
 * call before spg_prof_read(..., reset = 1) */
int spg_prof_tag(int kind, int it, int jt, int x, int y, int full);
/* per-(kernel instantiation, layer shape) totals of the instrumented launches: row j = keys[4j..4j+3] = {tag, N, K,
 * launches}, vals[2j..2j+1] = {milliseconds, FLOP}; returns the number of rows (<= max); call before
 * spg_prof_read(..., reset = 1) */
int spg_prof_read_shapes(int* keys, double* vals, int max);
/* Tuning knobs of the row-GEMM launches (process-global; the defaults are the production values).  key 0: 1 = launch
 * one workgroup per tile instead of the persistent chunk stream (A/B timing in tools/); key 3 (ONLY in a library built with
 * `make ATTRIBUTION=1`; the production build rejects it with -1): timing-attribution switches of the persistent forward
 * launches (bit 0 no output store, 1 no BatchNorm partials, 2 no pooling, 3 no epilogue at all, 4 A operand from L2, 5 no
 * main-loop barriers, bits 8-11 extra repetitions of the chunk loop) -- results are WRONG while it is non-zero; key 4: the BatchNorm finalize kernels slice their reduction over 8
 * workgroups (last-arrival combine) above this many partials (0 = default 512); key 5 (attribution build only): 1 = persistent
 * launches write one statistics partial per tile instead of one per workgroup; key 7: arithmetic of the wide (128-row-tile, full-tile) GEMMs of
 * spg_pointnet_forward / _backward: 0 = fp32 MFMA (default; the reference's arithmetic), 3 = split-bf16 (three bf16 MFMAs per
 * operand pair, ~2^-16 per product), 1 = bf16 operands; fp32 accumulation and fp32 tensors in every mode (tolerances:
 * tests/test_gpu_precision.py); key 8: 1 = run the GRU recurrence of spg_eccrnn_forward / _backward as one launch per
 * iteration instead of the persistent dataflow-synchronised launch (A/B timing and the equality test; the two forms give
 * bit-identical results), 2 = only the iteration-major form for more nodes than wavefronts (round 5) is off; key 9: 1 = the recurrent cell's parameter-gradient launches of spg_eccrnn_backward go to a
 * library-owned side stream next to the filter network's backward chain (experiment; measured slower, off by default).
 * key 10: 1 = train-mode BatchNorm statistics of spg_pointnet_forward go through per-workgroup partials and a finalize launch
 * per layer (the pre-round-3 path, still used with synchronised BatchNorm) instead of fixed-point slots finished by the
 * consuming GEMM.  key 11: 1 = no grouped launches (round 4: mutually independent few-row GEMMs / small reductions leave as
 * jobs of ONE kernel; the bodies are unchanged, results bit-identical -- A/B timing and the equality test).
 * key 12: 1 = few-row GEMMs with a reduction of >= 128 run in the split-K form (a 32 x 32 tile per workgroup, the four waves
 * split the reduction chunks; an experiment: measured no faster, off by default).
 * key 13: 1 = the masked (incomplete-tile) forward pipelines store scalars instead of vectors.  key 14: 1 = the backward of the
 * 64- / 128-input-channel convolutions runs as separate data-gradient and weight-gradient launches instead of the fused
 * one-pass kernel (same arithmetic per element, different summation order of dW and of the BatchNorm-backward sums).
 * key 15: 1 = spg_train_step runs the classifier and the cross entropy as separate launches instead of inside the one-launch
 * RNN-ECC recurrence (per element the same expressions; the dot products over the 32 state channels / the classes are
 * sequential fma chains instead of MFMA chunks: differences at fp32 round-off).  key 6: 1 = with the classifier inside the
 * recurrence, its weight / bias gradient leaves as a job of a grouped launch (+10 us) instead of being formed by service
 * workgroups of the persistent backward launch on the CUs the recurrence leaves idle.
 * key 16: 1 = spg_pointnet_backward hands the weight gradients of the pooled and of the first convolution to later launches as
 * LEAVES (round 5 experiment, OFF by default: slices of those launches -- row ranges of their split plan, same body, same plan:
 * results bit-identical -- travel next to the STN head's grouped launches instead of standing in front of the data gradients;
 * measured +29 us per step: a workgroup of the pooled layer's weight gradient runs ~60-90 us whatever the slice, the head's
 * launches are ~20 us each and already filled by the riding reductions -- DESIGN 4.16).
 * key 17: 1 = spg_pointnet_forward runs the first two convolutions of a segment (cloud -> 64 -> 64) as two row-GEMM launches
 * instead of the one-pass kernel of round 5 (spg_narrow.hip: first-layer statistics from the Gram matrix of the input, one
 * wavefront per block of 32 points through both layers; same MFMA order per element, the first layer's statistics exact instead
 * of accumulated from rounded outputs -- results agree at ~1e-6, tests/test_gpu_narrow.py).
 * key 18: 1 = spg_pointnet_backward runs the backward of a segment's first convolution as weight-gradient launch + xy data
 * gradient + spg_stn_dT instead of the one-pass kernel of round 5 (spg_narrow.hip: the layer's raw output is linear in the cloud,
 * so its part of the BatchNorm-backward formula collapses onto the Gram matrix -- one pass over the incoming gradient and the
 * cloud; results agree at fp32 round-off, tests/test_gpu_narrow.py).
 * key 19: 1 = the jobs of a grouped launch are ordered by workgroup length alone, as in round 4.  Default (round 5): the group's OWN
 * jobs -- what the next launch of the stream waits for -- take the first workgroup slots, riders / riding reductions / leaves
 * follow (bit-identical; -8.6 us per step: the job spans of an attribution build showed the FC head's data gradient starting 17 us
 * into a 42 us launch, behind 581 riding weight-gradient workgroups; bench.py --group-trace).
 * key 20: > 0 = a spin bound for the waits of the one-launch recurrences below the built-in one (tests only: tests/test_gpu_failsafe.py
 * forces a time-out with it); -1 = every wait reports a time-out at once, whether its data had arrived or not (deterministic on the
 * smallest graphs: tests/test_gpu_main.py); 0 (default) = the built-in bound.
 * key 21: 1 = spg_adam_clamp_step* ignore the time-out word of the one-launch recurrences (see spg_ecc_persistent_status); default 0:
 * the update is withheld while the word is non-zero.
 * key 22: 1 = the pooled convolution of a PointNet segment with 128 -> 256 channels keeps its separate weight-gradient and
 * data-gradient launches (+ finalize); default (round 6): the fused backward pair as two launches over the halves of its output
 * channels (spg_gemm.hip: spg_queue_bwdpair; 196 -> 167 us on the unit scene; results agree at fp32 round-off, tests/test_gpu_bwdpair.py).
 * Returns the previous value, -1 for an unknown key. */
int spg_tune(int key, int value);
/* ------------------------------------------------------------------------------------------------
 * One training step's forward AND backward in ONE call (round 4): CloudEmbedder.run -> model.ecc (RNN-ECC module +
 * classifier) -> weighted cross entropy -> backward -> bw_hook, i.e. learning/main.py:199-208 for the standard model
 * (`gru_R.../lstm_R...` followed by `f_K`), with every gradient written to the caller's buffers (the flat gradient arena).
 * Same expressions as the module-level calls it replaces (spg_pointnet_forward_ext, spg_gather_rows, spg_eccrnn_forward,
 * spg_linear_fwd, spg_cross_entropy_*, spg_linear_backward, spg_eccrnn_backward, spg_pointnet_backward_ext); with spg_tune key
 * 15 = 1 the same kernels and bit-identical results.  By default the classifier and the cross entropy are computed inside the
 * one-launch GRU recurrence, per node by the wavefront that owns it (the classifier's weight gradient by service workgroups of
 * the backward recurrence's launch): no launches of their own, results equal at fp32 round-off (other summation order of the
 * 32-channel dot products; tests/test_gpu_fused.py).  Otherwise what the call adds is the ORDER of launches: inside one call the library knows the whole step, so
 * the filter network's forward (needs only the superedge features) leaves next to PointNet's few-row launches and the tail of
 * the RNN-ECC backward (cell and filter-network parameter gradients, needed by nobody before the optimiser) next to
 * PointNet's backward, as jobs of the same grouped launches instead of ~12 latency-bound launches of their own; and the host
 * enqueues a step with one ctypes call instead of ~15 autograd nodes.  The clamp + Adam update stays a separate call
 * (spg_adam_clamp_step_scaled): the data-parallel all-reduce sits between the two.
 * Buffers: everything is caller-allocated; workspace sizes from the *_workspace_bytes queries of the two networks. */
typedef struct spg_step_args {
  /* PointNet over the B embeddable superpoints (learning/pointnet.py:138-180) */
  const spg_pointnet_cfg* ptn_cfg;
  int B, bn_update_times;                 /* bn_update_times = 2 with ptn_mem_monger (the reference's forward + re-forward) */
  const float* clouds;                    /* [B, nfeat, npts] */
  const float* clouds_global;             /* [B, nfeat_global] */
  const void* const* ptn_params;          /* 6 pointers per layer (spg_pointnet_forward) */
  void* const* ptn_grads;                 /* 6 pointers per layer (spg_pointnet_backward) */
  void* ptn_ws; void* ptn_bwd_ws;
  float* emb;                             /* out [B, nf]: PointNet embeddings */
  float* grad_emb;                        /* scratch [B, nf] */
  /* scatter to all N superpoints: descriptors[i] = emb[slot_of_row[i]] (zero row where slot < 0, pointnet.py:177-179) */
  int N, nf;
  const int64_t* slot_of_row;             /* [N] */
  const int64_t* idx_valid;               /* [B] rows of the valid superpoints */
  float* desc;                            /* out [N, nf] */
  float* grad_desc;                       /* scratch [N, nf] */
  /* RNN-ECC module (learning/modules.py:152-183) */
  const spg_eccrnn_cfg* ecc_cfg;
  int E;
  const void* graph_ws;
  const float* edgefeats;                 /* [E, fnet_widths[0]] in the batch's edge order */
  const void* const* ecc_params;
  void* const* ecc_grads;
  void* ecc_ws; void* ecc_bwd_ws;
  float* ecc_out;                         /* out [N, nout] (nout = 32 or 32 * (R + 1) with cat_all) */
  float* grad_ecc_out;                    /* scratch [N, nout] */
  /* classifier Linear(nout -> n_classes) (learning/graphnet.py:47-49) */
  int nout, n_classes;
  const float* cls_W; const float* cls_b; /* [n_classes, nout], [n_classes] or null */
  float* cls_dW; float* cls_db;
  float* cls_work;                        /* >= spg_linear_wgrad_bias_work_floats(N, n_classes, nout) floats */
  float* logits;                          /* out [N, n_classes] */
  float* grad_logits;                     /* scratch [N, n_classes] */
  /* weighted cross entropy (learning/main.py:205) */
  const int64_t* target;                  /* [N] class index or ignore_index */
  const float* class_weight;              /* [n_classes] or null */
  int64_t ignore_index;
  int reduction_mean;                     /* 1: mean over the labelled rows (single process); 0: sum (data parallel) */
  float* loss_buf;                        /* out [N + 2]: log-sum-exp per row | loss | sum of the labelled rows' class weights */
  /* in: 1 = the BatchNorm statistics slots inside ptn_ws are zero -- true when the previous call that used this ptn_ws was a
   * spg_train_step that returned 0 (every step leaves them zero: the clearing rides with its last launch) and nothing else has
   * written ptn_ws since; 0 (always safe): the step clears them itself first (one more launch) */
  int ptn_slots_clean;
} spg_step_args;
int spg_train_step(const spg_step_args* args, void* stream);
/* The forward of the same model in INFERENCE mode (BatchNorm running statistics) as one call: the evaluation loop's body
 * (learning/main.py:256-262: ptnCloudEmbedder.run + model.ecc under model.eval()).  Same kernels and results as the module-level
 * calls.  Reads of spg_step_args: the forward inputs, parameters, ptn_ws / ecc_ws (sizes from the *_workspace_bytes queries with
 * training = 0, or larger), emb, the scatter tables (desc: written only by the per-iteration RNN-ECC fallback), ecc_out, the
 * classifier and logits; the gradient / loss fields are ignored. */
int spg_infer_step(const spg_step_args* args, void* stream);
/* loss, log-sum-exp, normaliser AND the gradient wrt the logits (for d loss = 1) in one single-workgroup launch */
int spg_cross_entropy_fwd_bwd(const float* logits, const int64_t* target, const float* weight, int N, int C, int64_t ignore_index,
                              int reduction_mean, float* loss, float* lse, float* wsum, float* grad_logits, void* stream);

/* Number of bounded-spin time-outs the persistent RNN-ECC launches of the current device have raised so far (0 in a correct
 * run; a wave that waits too long for a neighbour's state gives up instead of hanging the GPU).  Synchronises the device. */
int spg_ecc_persistent_errors(void);
/* The same count, and the error word is cleared.  Production callers (learning/main.py: once per epoch and before every
 * checkpoint, where the host synchronises anyway) raise when it is non-zero: the affected launches carried on with stale
 * neighbour states, i.e. the ECC outputs / gradients of those steps are wrong (spg_tune key 8 = 1 selects the
 * per-iteration kernels, which cannot time out). */
int spg_ecc_persistent_errors_clear(void);
/* Fail-safe of the optimiser step (round 6).  The error word above is STICKY and lives on the device; spg_adam_clamp_step /
 * spg_adam_clamp_step_scaled read it inside their launch and WITHHOLD the update while it is non-zero -- parameters and both moment
 * buffers stay bit-identical -- counting the withheld launches.  No host synchronisation: a step whose recurrence timed out, and every
 * step after it, changes nothing until the host has looked.  This call returns both counts (*errors: time-outs, *withheld: update
 * launches skipped) and, with clear != 0, zeroes them; one blocking 16-byte copy each way, so call it where the host may wait
 * (learning/main.py: every --ecc_check_every steps, at the end of an epoch, before a checkpoint).  The caller then switches to the
 * per-iteration kernels (spg_tune key 8 = 1), takes `withheld` off its Adam step counter and repeats the batch(es) it still holds.
 * The reference's only guard on this path is the NaN-loss check of learning/main.py:367. */
int spg_ecc_persistent_status(int* errors, int* withheld, int clear);
int spg_prof_read_tag(int tag, double* ms, long* launches, double* flops);

/* (new, tools only) Attribution builds of the library (make ATTRIBUTION=1): per-job time spans of the grouped launches -- buf: device
 * memory [max_launches][16][2] unsigned long long, pre-filled (starts ~0, ends 0); null switches the recording off.  A production build
 * returns -1.  spg_group_trace_read: the host-side log {njobs, heavy, njobs x {kind, variant, gx, gy, gz, weight}} per traced launch;
 * returns the number of ints available.  (bench.py --group-trace) */
int spg_group_trace(void* buf, int max_launches);
int spg_group_trace_read(int* out, int max);

#ifdef __cplusplus
}
#endif
#endif /* SPG_HIP_H */
