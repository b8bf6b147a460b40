// One training step's forward + backward in one C-ABI call (include/spg_hip.h: spg_train_step): the launch ORDER of a step
// whose parts the library sees together.  reference: learning/main.py:199-208 (the trainer's loop body), CloudEmbedder.run_full
// (learning/pointnet.py:155-180), GraphNetwork.forward (learning/graphnet.py:95-98).
//
// What is gained over the module-level calls (same kernels, same results):
//   * the filter-generating network of the RNN-ECC module needs only the superedge features: its four dependent few-row
//     GEMMs are registered as RIDERS (spg_gemm.h) and leave as extra jobs of PointNet's own few-row launches (the STN / FC
//     heads) -- ~45 us of latency-bound launches vanish from the stream;
//   * the tail of the RNN-ECC backward (the cell's and the filter network's parameter gradients: six dependent grouped
//     launches, ~120 us) is needed by nobody before the optimiser step: it rides next to PointNet's backward the same way,
//     and the split partials of all three modules are summed by ONE batched reduction at the very end;
//   * cross entropy forward + backward are one launch;
//   * the host enqueues the step with one call.
#include "../../include/spg_hip.h"
#include "spg_ecc.h"
#include "spg_gemm.h"

int spg_eccrnn_forward_phase(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* h0, const float* edgefeats,
                             const void* const* params, float* out, void* workspace, int training, int bn_update_times, void* stream,
                             int phase, const SpgEccScatter* sc, const SpgEccHead* head, bool* head_done);
int spg_eccrnn_backward_phase(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* edgefeats,
                              const void* const* params, const float* grad_out, float* grad_h0, void* const* grads,
                              void* workspace, void* bwd_workspace, void* stream, int phase, const SpgEccScatter* sc,
                              SpgStage extra_leaf, const SpgEccHead* head);
int spg_linear_wgrad_bias_queue_deferred(const float* dY, long lddy, const float* X, long ldx, int M, int N, int K, float* dW,
                                         float* dbias, float* work, hipStream_t st);
int spg_linear_backward_deferred(const float* dY, long lddy, const float* X, long ldx, const float* W, int M, int N, int K, float* dX,
                                 long lddx, float* dW, float* dbias, float* work, hipStream_t st);

void spg_pointnet_set_step_flags(bool clean, bool clear_at_end);

namespace {
// whatever happens inside the step, no rider / deferred reduction survives the call (their buffers belong to the caller)
struct StepGuard {
  ~StepGuard() { spg_riders_clear(); spg_reduce_deferred_clear(); spg_pointnet_set_step_flags(false, false); }
};
}  // namespace

extern "C" int spg_train_step(const spg_step_args* a, void* stream) {
  SPG_CHECK_ARG(a != nullptr, "null arguments");
  SPG_CHECK_ARG(a->ptn_cfg && a->ecc_cfg && a->B > 0 && a->N > 0 && a->E >= 0, "cfg / sizes");
  SPG_CHECK_ARG(a->clouds && a->ptn_params && a->ptn_grads && a->ptn_ws && a->ptn_bwd_ws && a->emb && a->grad_emb, "PointNet buffers");
  SPG_CHECK_ARG(a->slot_of_row && a->idx_valid && a->desc && a->grad_desc && a->nf == a->ecc_cfg->nc, "scatter buffers / embedding width");
  SPG_CHECK_ARG(a->graph_ws && a->ecc_params && a->ecc_grads && a->ecc_ws && a->ecc_bwd_ws && a->ecc_out && a->grad_ecc_out, "RNN-ECC buffers");
  SPG_CHECK_ARG(a->cls_W && a->cls_dW && a->cls_work && a->logits && a->grad_logits && a->nout > 0 && (a->nout & 3) == 0 && a->n_classes > 0, "classifier buffers");
  SPG_CHECK_ARG(a->target && a->loss_buf, "loss buffers");
  SPG_CHECK_ARG(spg_riders_pending() == 0, "a rider chain is already registered on this thread");
  hipStream_t st = (hipStream_t)stream;
  const int N = a->N, E = a->E, B = a->B;
  StepGuard guard;
  spg_reduce_deferred_clear();
  // ---------------- forward ----------------
  // the filter network's layers travel with PointNet's few-row launches
  SpgEccScatter sc;
  sc.emb = a->emb; sc.slot_of_row = a->slot_of_row; sc.idx_valid = a->idx_valid; sc.desc = a->desc; sc.grad_emb = a->grad_emb; sc.B = B;
  SPG_TRY(spg_eccrnn_forward_phase(a->ecc_cfg, N, E, a->graph_ws, nullptr, a->edgefeats, a->ecc_params, nullptr, a->ecc_ws, 1, 1, stream, 1, nullptr, nullptr, nullptr));
  // BatchNorm statistics slots inside ptn_ws: every step leaves them ZERO (the clearing rides with the final reduction below), so
  // a caller that runs step after step on the same workspace saves the memset launch in front of every forward
  spg_pointnet_set_step_flags(a->ptn_slots_clean != 0, false);
  SPG_TRY(spg_pointnet_forward_ext(a->ptn_cfg, B, a->clouds, a->clouds_global, nullptr, a->ptn_params, a->emb, a->ptn_ws, 1,
                                   a->bn_update_times, stream));
  spg_pointnet_set_step_flags(false, false);
  SPG_TRY(spg_riders_drain(st));
  // (the embedding scatter is read in place by the one-launch recurrence; its per-iteration fallback materialises a->desc)
  // the classifier and the cross entropy behind the last iteration: computed per node by the wavefront that owns it, inside the
  // one-launch recurrence (SpgEccHead, spg_ecc.h) -- logits, loss, d loss / d logits and d loss / d h^R leave that launch; the
  // classifier's own parameter gradients are a leaf and ride with the tail of the RNN-ECC backward.  Served: <= 32 classes whose
  // rows fit SPG_PX_HEAD_LDS bytes of LDS; spg_tune key 15 = 1, LSTM or the per-iteration fallback: separate launches.
  SpgEccHead head; memset(&head, 0, sizeof(head));
  const bool try_head = !spg_tune_get(SPG_TUNE_NO_ECC_HEAD) && a->n_classes <= SPG_PX_HEAD_MAXC;
  if (try_head) {
    head.W = a->cls_W; head.b = a->cls_b; head.target = a->target; head.class_weight = a->class_weight;
    head.ignore_index = a->ignore_index; head.C = a->n_classes; head.reduction_mean = a->reduction_mean; head.N = N; head.nin = a->nout;
    head.logits = a->logits; head.grad_logits = a->grad_logits; head.grad_out = a->grad_ecc_out;
    head.lse = a->loss_buf; head.loss = a->loss_buf + N; head.wsum = a->loss_buf + N + 1; head.dW = a->cls_dW; head.db = a->cls_db;
  }
  bool head_done = false;
  SPG_TRY(spg_eccrnn_forward_phase(a->ecc_cfg, N, E, a->graph_ws, nullptr, a->edgefeats, a->ecc_params, a->ecc_out, a->ecc_ws, 1, 1, stream, 2, &sc,
                                   try_head ? &head : nullptr, &head_done));
  SpgStage cls_leaf;
  if (!head_done) {
    SPG_TRY(spg_linear_fwd(a->ecc_out, a->nout, N, a->nout, a->cls_W, a->cls_b, a->n_classes, nullptr, nullptr, 0, a->logits, a->n_classes, stream));
    SPG_TRY(spg_cross_entropy_fwd_bwd(a->logits, a->target, a->class_weight, N, a->n_classes, a->ignore_index, a->reduction_mean,
                                      a->loss_buf + N, a->loss_buf, a->loss_buf + N + 1, a->grad_logits, stream));
    // ---------------- backward ----------------
    SPG_TRY(spg_linear_backward_deferred(a->grad_logits, a->n_classes, a->ecc_out, a->nout, a->cls_W, N, a->n_classes, a->nout,
                                         a->grad_ecc_out, a->nout, a->cls_dW, a->cls_db, a->cls_work, st));
  } else {
    const spg_step_args args = *a;      // (the stage runs inside this call; a copy keeps the closure self-contained)
    cls_leaf = [args](hipStream_t s2) -> int {
      return spg_linear_wgrad_bias_queue_deferred(args.grad_logits, args.n_classes, args.ecc_out, args.nout, args.N, args.n_classes, args.nout,
                                                  args.cls_dW, args.cls_db, args.cls_work, s2);
    };
  }
  // through the recurrence; its tail rides with PointNet's backward
  SPG_TRY(spg_eccrnn_backward_phase(a->ecc_cfg, N, E, a->graph_ws, a->edgefeats, a->ecc_params, a->grad_ecc_out, a->grad_desc, a->ecc_grads,
                                    a->ecc_ws, a->ecc_bwd_ws, stream, 1, &sc, std::move(cls_leaf), head_done ? &head : nullptr));
  spg_pointnet_set_step_flags(false, true);
  SPG_TRY(spg_pointnet_backward_ext(a->ptn_cfg, B, a->clouds, a->clouds_global, nullptr, a->ptn_params, a->grad_emb, a->ptn_grads, nullptr,
                                    nullptr, a->ptn_ws, a->ptn_bwd_ws, stream));
  spg_pointnet_set_step_flags(false, false);
  SPG_TRY(spg_riders_drain(st));
  return spg_flush_deferred_reduce(st);
}

// The forward of the same model in INFERENCE mode (BatchNorm running statistics, no gradients kept) as one call:
// CloudEmbedder.run -> model.ecc under model.eval() / torch.no_grad() (learning/main.py:256-262, the evaluation loop's body).
// Same kernels and results as the module-level calls; the host enqueues it once instead of walking ~10 modules.  Uses of
// spg_step_args: the forward inputs / parameters / workspaces (ptn_ws, ecc_ws sized for training = 0 or larger), emb, the
// scatter tables, ecc_out, the classifier and logits; everything about gradients and the loss is ignored.
extern "C" int spg_infer_step(const spg_step_args* a, void* stream) {
  SPG_CHECK_ARG(a != nullptr, "null arguments");
  SPG_CHECK_ARG(a->ptn_cfg && a->ecc_cfg && a->B > 0 && a->N > 0 && a->E >= 0, "cfg / sizes");
  SPG_CHECK_ARG(a->clouds && a->ptn_params && a->ptn_ws && a->emb, "PointNet buffers");
  SPG_CHECK_ARG(a->slot_of_row && a->idx_valid && a->desc && a->nf == a->ecc_cfg->nc, "scatter buffers / embedding width");
  SPG_CHECK_ARG(a->graph_ws && a->ecc_params && a->ecc_ws && a->ecc_out, "RNN-ECC buffers");
  SPG_CHECK_ARG(a->cls_W && a->logits && a->nout > 0 && (a->nout & 3) == 0 && a->n_classes > 0, "classifier buffers");
  SPG_CHECK_ARG(spg_riders_pending() == 0, "a rider chain is already registered on this thread");
  SpgEccScatter sc;
  sc.emb = a->emb; sc.slot_of_row = a->slot_of_row; sc.idx_valid = a->idx_valid; sc.desc = a->desc; sc.grad_emb = nullptr; sc.B = a->B;
  SPG_TRY(spg_pointnet_forward_ext(a->ptn_cfg, a->B, a->clouds, a->clouds_global, nullptr, a->ptn_params, a->emb, a->ptn_ws, 0, 1, stream));
  SPG_TRY(spg_eccrnn_forward_phase(a->ecc_cfg, a->N, a->E, a->graph_ws, nullptr, a->edgefeats, a->ecc_params, a->ecc_out, a->ecc_ws, 0, 1, stream, 0, &sc,
                                   nullptr, nullptr));
  return spg_linear_fwd(a->ecc_out, a->nout, a->N, a->nout, a->cls_W, a->cls_b, a->n_classes, nullptr, nullptr, 0, a->logits, a->n_classes, stream);
}
