// RNN-ECC module (reference: RNNGraphConvModule.forward, learning/modules.py:152-183): filter-generating MLP on
// the superedge features (learning/graphnet.py:17-34), then nrepeats x {ECC aggregate, GRUCellEx}.
// Host-side orchestration of the forward / backward kernel sequences and the workspace layout.
//
// HBM layout: every per-node quantity that exists once per iteration is stored as [N][R+1][C]
// (states h^r, aggregates, and in the backward the gate gradients), so that "all iterations" is a plain
// row-major matrix with (R+1)*N rows for the deferred weight-gradient GEMMs, and a node's history is
// contiguous for the per-edge filter gradient (summed over the iterations in registers, written once).
#include "../../include/spg_hip.h"
#include "spg_ecc.h"
#include "spg_gemm.h"
#include <memory>
#include <vector>

namespace {

struct FLayer {
  int cin = 0, cout = 0;
  bool bn = false, relu = false;
  const float *W = nullptr, *b = nullptr, *gamma = nullptr, *beta = nullptr;
  float *rm = nullptr, *rv = nullptr;
  float* y = nullptr;
  float *mean = nullptr, *rstd = nullptr, *s = nullptr, *t = nullptr;
  unsigned long long *slots = nullptr, *slots_bwd = nullptr;   // train mode: fixed-point statistics slots (SpgBnFold, spg_gemm.h)
  float *dW = nullptr, *db = nullptr, *dgamma = nullptr, *dbeta = nullptr;
};

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct Plan {
  spg_eccrnn_cfg cfg;
  int N = 0, E = 0, R = 0, nout = 0;
  long ldS = 0;                 // (R+1)*32
  int GW = 96;                  // gate rows of the cell: 96 (GRU) / 128 (LSTM)
  bool lstm = false;
  bool training = false;
  std::vector<FLayer> F;
  bool fold = false;            // train mode: the BatchNorm statistics travel as fixed-point slots from producer to consumer GEMM
  bool px = false;              // the persistent (one launch for all iterations) GRU recurrence is applicable: rounds in `groups`
  SpgPxGroups groups;
  SpgGruParams gru;
  float *states = nullptr, *agg = nullptr, *stat = nullptr, *stat_cnt = nullptr;
  float* cells = nullptr;       // LSTM cell states c^r, laid out like `states`
  float* fsave = nullptr;       // persistent GRU recurrence, training: forward internals kept for the backward
  unsigned* fsave_tag = nullptr;
  float* cell_grads[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t bytes = 0;
};

int make_plan(const spg_eccrnn_cfg* cfg, int N, int E, int training, void* ws, const void* const* params, Plan& pl) {
  SPG_CHECK_ARG(cfg != nullptr && N > 0 && E >= 0, "cfg / N / E");
  const spg_eccrnn_cfg& c = *cfg;
  SPG_CHECK_ARG(c.nc == 32, "the fused RNN-ECC path needs 32 channels");
  SPG_CHECK_ARG(c.nrepeats >= 1, "nrepeats >= 1");
  SPG_CHECK_ARG(c.n_fnet >= 1 && c.n_fnet <= SPG_MAX_LAYERS, "n_fnet");
  SPG_CHECK_ARG(c.bnidx < c.n_fnet - 1 || c.bnidx < 0, "BatchNorm after the last filter layer is not supported");
  SPG_CHECK_ARG(c.cell == SPG_CELL_GRU || c.cell == SPG_CELL_LSTM, "cell must be 0 (GRU) or 1 (LSTM)");
  pl.cfg = c; pl.N = N; pl.E = E; pl.R = c.nrepeats; pl.training = training != 0;
  pl.lstm = c.cell == SPG_CELL_LSTM; pl.GW = pl.lstm ? 128 : 96;
  pl.ldS = (long)(pl.R + 1) * 32;
  pl.nout = c.matrix ? 1024 : 32;
  SPG_CHECK_ARG(c.fnet_widths[c.n_fnet] == pl.nout, "filter network output width must be nc*nc (matrix) or nc (vector)");
  Carver cv(ws);
  pl.F.assign(c.n_fnet, FLayer());
  int cmax = 4;
  for (int i = 0; i < c.n_fnet; ++i) {
    FLayer& l = pl.F[i];
    l.cin = c.fnet_widths[i]; l.cout = c.fnet_widths[i + 1];
    l.bn = (c.bnidx == i); l.relu = i < c.n_fnet - 1;
    cmax = l.cout > cmax ? l.cout : cmax;
    if (params) {
      const void* const* g = params + 6 * i;
      l.W = (const float*)g[0]; l.b = (const float*)g[1]; l.gamma = (const float*)g[2]; l.beta = (const float*)g[3];
      l.rm = (float*)g[4]; l.rv = (float*)g[5];
      SPG_CHECK_ARG(l.W != nullptr, "missing filter-network weight");
      SPG_CHECK_ARG(!l.bn || (l.rm && l.rv), "missing BatchNorm running statistics");
    }
    l.y = cv.take<float>((size_t)(E > 0 ? E : 1) * l.cout);
    if (l.bn) {
      l.mean = cv.take<float>(l.cout); l.rstd = cv.take<float>(l.cout);
      l.s = cv.take<float>(l.cout); l.t = cv.take<float>(l.cout);
      if (pl.training) {
        l.slots = cv.take<unsigned long long>(spg_fold_slot_words(l.cout));
        l.slots_bwd = cv.take<unsigned long long>(spg_fold_slot_words(l.cout));
      }
    }
  }
  memset(&pl.gru, 0, sizeof(pl.gru));
  if (params) {
    const void* const* g = params + 6 * c.n_fnet;
    pl.gru.w_ih = (const float*)g[0]; pl.gru.w_hh = (const float*)g[1];
    pl.gru.b_ih = (const float*)g[2]; pl.gru.b_hh = (const float*)g[3];
    pl.gru.w_ig = (const float*)g[4]; pl.gru.b_ig = (const float*)g[5];
    SPG_CHECK_ARG(pl.gru.w_ih && pl.gru.w_hh && pl.gru.b_ih && pl.gru.b_hh, "missing RNN cell parameters");
    SPG_CHECK_ARG(!c.ingate || (pl.gru.w_ig && pl.gru.b_ig), "missing input-gate parameters");
  }
  pl.gru.layernorm = c.layernorm; pl.gru.ingate = c.ingate;
  pl.states = cv.take<float>((size_t)N * pl.ldS);
  pl.agg = cv.take<float>((size_t)N * pl.ldS);
  if (pl.lstm) pl.cells = cv.take<float>((size_t)N * pl.ldS);
  // persistent GRU recurrence in training: the forward keeps the cell's internals of every (node, iteration) for the backward
  // (12 x 64 floats each: 31 MB per 1000 nodes x 10 iterations) instead of the backward recomputing them on its critical path
  pl.fsave = nullptr; pl.fsave_tag = nullptr;
  pl.px = !pl.lstm && c.n_parts >= 0 && c.n_parts <= SPG_MAX_PARTS && spg_px_plan_groups(N, c.n_parts, c.part_ptr, &pl.groups);
  // (more nodes than wavefronts -- one round above SPG_PX_MAX_NODES nodes: the backward recomputes the cell instead, spg_ecc.hip)
  if (pl.px && pl.training && !spg_px_is_multi(pl.groups.n, N)) {
    pl.fsave_tag = cv.take<unsigned>(64);
    pl.fsave = cv.take<float>((size_t)N * pl.R * SPG_PX_SAVE_F * 64);
  }
  pl.stat = cv.take<float>((size_t)spg_cdiv(E > 0 ? E : 1, SPG_FC_ROWS) * 2 * cmax);
  pl.stat_cnt = cv.take<float>((size_t)spg_cdiv(E > 0 ? E : 1, SPG_FC_ROWS) + 64);
  pl.bytes = cv.off + 256;
  return 0;
}

SpgOperand op_ident(const float* X, long ld) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_IDENT; o.X = X; o.ld = ld;
  return o;
}

SpgOperand fnet_input(const Plan& pl, int i, const float* edgefeats) {
  if (i == 0) return op_ident(edgefeats, pl.F[0].cin);
  const FLayer& p = pl.F[i - 1];
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_AFFINE; o.X = p.y; o.ld = p.cout; o.relu = p.relu ? 1 : 0; o.n_affine = p.cout;
  if (p.bn) { o.c0 = p.s; o.c1 = p.t; }
  return o;
}

SpgOperand op_bnbwd(const float* dz, const float* y, long ld, const float* consts, int C) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_BNBWD; o.X = dz; o.X2 = y; o.ld = ld;
  o.c0 = consts; o.c1 = consts + C; o.c2 = consts + 2 * C; o.c3 = consts + 3 * C;
  return o;
}

int zero_async(void* p, size_t bytes, hipStream_t st) {
  if (p == nullptr || bytes == 0) return 0;
  hipError_t e = hipMemsetAsync(p, 0, bytes, st);
  if (e != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

struct BwdScratch {
  float *G = nullptr, *dgi = nullptr, *dgh = nullptr, *dui = nullptr, *duh = nullptr, *dpre = nullptr, *xg = nullptr;
  float *dcdir = nullptr;
  float *dhdir = nullptr, *dWts = nullptr, *dzA = nullptr, *dzB = nullptr, *Wt = nullptr, *consts = nullptr;
  float *work = nullptr, *stat = nullptr;
  float* work2 = nullptr;   // reduction arena of the recurrent cell's parameter gradients (they may run on the side stream)
  size_t work_floats = 0, work2_floats = 0;
  size_t zero_bytes = 0;   // the leading region [G .. xg] must be zero-initialised
  size_t bytes = 0;
};

void carve_bwd(const Plan& pl, void* ws, BwdScratch& s) {
  Carver cv(ws);
  const size_t rows = (size_t)pl.N * (pl.R + 1);
  s.G = cv.take<float>(rows * 32);
  s.dgi = cv.take<float>(rows * pl.GW); s.dgh = cv.take<float>(rows * pl.GW);
  if (!pl.lstm) { s.dui = cv.take<float>(rows * 96); s.duh = cv.take<float>(rows * 96); }
  s.dpre = cv.take<float>(rows * 32); s.xg = cv.take<float>(rows * 32);
  s.zero_bytes = cv.off;
  s.dhdir = cv.take<float>((size_t)pl.N * 32);
  if (pl.lstm) s.dcdir = cv.take<float>((size_t)pl.N * 32);
  const size_t Er = pl.E > 0 ? pl.E : 1;
  s.dWts = cv.take<float>(Er * pl.nout);
  int cmax = 4;
  size_t wmax = 96 * 32, workmax = 16;
  for (const FLayer& l : pl.F) {
    cmax = l.cout > cmax ? l.cout : cmax; cmax = l.cin > cmax ? l.cin : cmax;
    wmax = (size_t)l.cin * l.cout > wmax ? (size_t)l.cin * l.cout : wmax;
    workmax += ((spg_wgrad_workspace_floats(Er, l.cout, l.cin) + 63) & ~(size_t)63) + 64 * (size_t)l.cout + 128 +
               ((spg_wgrad_colsum_floats((long)Er, l.cout, l.cin) + 63) & ~(size_t)63);
  }
  // GRU: three weight gradients + three bias column sums over all (node, iteration) rows
  const size_t work2max = 16 + 3 * (((spg_wgrad_workspace_floats((long)rows, pl.GW, 32) + 63) & ~(size_t)63) + 64 * (size_t)pl.GW + 128 +
                                    ((spg_wgrad_colsum_floats((long)rows, pl.GW, 32) + 63) & ~(size_t)63));
  int hmax = 4;   // widest hidden activation
  for (int i = 0; i + 1 < (int)pl.F.size(); ++i) hmax = pl.F[i].cout > hmax ? pl.F[i].cout : hmax;
  s.dzA = cv.take<float>(Er * hmax); s.dzB = cv.take<float>(Er * hmax);
  s.consts = cv.take<float>((size_t)4 * cmax);
  s.work = cv.take<float>(workmax); s.work_floats = workmax;
  s.work2 = cv.take<float>(work2max); s.work2_floats = work2max;
  s.stat = cv.take<float>((size_t)spg_cdiv(Er, SPG_FC_ROWS) * 2 * cmax);
  s.bytes = cv.off + 256;
}

}  // namespace

extern "C" size_t spg_eccrnn_workspace_bytes(const spg_eccrnn_cfg* cfg, int N, int E, int training) {
  Plan pl;
  if (make_plan(cfg, N, E, training, nullptr, nullptr, pl) != 0) return 0;
  return pl.bytes;
}

// train-mode BatchNorm of the filter network without finalize launches, like PointNet's (spg_gemm.h: SpgBnFold): not with
// synchronised BatchNorm (the ranks' all-reduce sits between producer and consumer), not beyond the slots' capacity
static bool fnet_fold(const Plan& pl) {
  return pl.training && !spg_sync_bn_active() && !spg_tune_get(SPG_TUNE_NO_BN_FOLD) && (long)spg_cdiv(pl.E > 0 ? pl.E : 1, SPG_FC_ROWS) * spg_slot_sync_world() <= SPG_FOLD_MAX_CONTRIBUTIONS;
}

// ---- filter-generating network (once per forward, shared by all iterations): one stage per layer; each stage issues its
//      launches into the group that is open when it runs (spg_gemm.h) ----
static void fnet_forward_stages(const Plan& pl0, const float* edgefeats, int bn_update_times, std::vector<SpgStage>& out) {
  auto plan = std::make_shared<Plan>(pl0);
  for (int i = 0; i < (int)plan->F.size(); ++i) {
    out.push_back([plan, i, edgefeats, bn_update_times](hipStream_t st) -> int {
      const Plan& pl = *plan;
      const FLayer& l = pl.F[i];
      if (i == 0 && pl.fold)                 // the statistics slots of both directions: cleared launches before their first use
        for (const FLayer& f : pl.F)
          if (f.bn) {
            // BatchNorm behind the FIRST layer (--fnet_bnidx 0 is legal, learning/graphnet.py:17-37): this very stage's GEMM is the
            // slots' first producer, so a zero job in the same group would race with it -- clear them in stream order instead
            if (&f == &pl.F[0]) SPG_TRY(zero_async(f.slots, sizeof(float) * 2 * spg_fold_slot_words(f.cout), st));
            else SPG_TRY(spg_group_zero(reinterpret_cast<float*>(f.slots), 2 * spg_fold_slot_words(f.cout), st));
            SPG_TRY(spg_group_zero(reinterpret_cast<float*>(f.slots_bwd), 2 * spg_fold_slot_words(f.cout), st));
          }
      SpgGemmParams g; memset(&g, 0, sizeof(g));
      g.a = fnet_input(pl, i, edgefeats);
      g.W = l.W; g.ldw = l.cin; g.bias = l.b; g.M = pl.E; g.N = l.cout; g.K = l.cin; g.rows_per_tile = SPG_FC_ROWS;
      g.epi = SPG_EPI_FWD; g.Y = l.y; g.ldy = l.cout;
      g.stat = (l.bn && pl.training) ? pl.stat : nullptr; g.stat_cnt = pl.stat_cnt;
      if (l.bn && pl.fold) { g.stat = nullptr; g.stat_slots = l.slots; }
      if (i > 0 && pl.F[i - 1].bn && pl.fold) {      // the producer's statistics are finished in this launch's prologue
        const FLayer& p = pl.F[i - 1];
        SpgBnFold f; memset(&f, 0, sizeof(f));
        f.slots = p.slots; f.C = p.cout; f.update_times = bn_update_times; f.momentum = pl.cfg.bn_momentum; f.eps = pl.cfg.bn_eps;
        f.count = (double)pl.E; f.gamma = p.gamma; f.beta = p.beta; f.rm = p.rm; f.rv = p.rv;
        f.mean = p.mean; f.rstd = p.rstd; f.s = p.s; f.t = p.t;
        g.fold = f;
      }
      int nparts = 0;
      {
        // a finalize launch behind the GEMM (synchronised BatchNorm / fold switched off): the GEMM must not wait in a group
        SpgGroupBypass direct(l.bn && !pl.fold && pl.training);
        SPG_TRY(spg_launch_gemm(g, st, &nparts));
      }
      if (l.bn && !pl.fold) {
        if (pl.training)
          SPG_TRY(spg_launch_bn_finalize(pl.stat, pl.stat_cnt, nparts, pl.E, l.cout, l.gamma, l.beta, l.rm, l.rv,
                                         pl.cfg.bn_momentum, pl.cfg.bn_eps, bn_update_times, l.mean, l.rstd, l.s, l.t, nullptr, st));
        else
          SPG_TRY(spg_launch_bn_eval(l.cout, l.gamma, l.beta, l.rm, l.rv, pl.cfg.bn_eps, l.s, l.t, st));
      }
      return 0;
    });
  }
}

// head (optional, spg_train_step): classifier + cross entropy behind the last iteration, inside the persistent launch
// (SpgEccHead, spg_ecc.h); *head_done says whether that launch took it -- the caller runs the separate launches otherwise
static int eccrnn_recurrent_forward(Plan& pl, const void* graph_ws, const float* h0, float* out, hipStream_t st,
                                    const SpgEccScatter* sc, const SpgEccHead* head = nullptr, bool* head_done = nullptr) {
  if (head_done != nullptr) *head_done = false;
  const int N = pl.N, E = pl.E;
  SpgGraph gr = spg_graph_view(graph_ws, N, E);
  const int64_t* h0_rows = nullptr;
  if (sc != nullptr) { h0 = sc->emb; h0_rows = sc->slot_of_row; }      // the embedding scatter read in place (persistent launch)
  if (pl.px) {      // GRU: all iterations in one dataflow-synchronised launch, whole scenes in rounds of <= 2048 nodes (spg_ecc.hip)
    SpgEccPersistFwd q; memset(&q, 0, sizeof(q));
    q.groups = pl.groups;
    q.g = gr; q.W = pl.F.back().y; q.matrix = pl.cfg.matrix; q.R = pl.R; q.h0 = h0; q.h0_rows = h0_rows;
    q.states = pl.states; q.ldS = pl.ldS; q.agg = pl.training ? pl.agg : nullptr;
    q.out = out; q.cat_all = pl.cfg.cat_all; q.ldo = pl.cfg.cat_all ? pl.ldS : 32; q.gru = pl.gru;
    q.fsave = pl.fsave; q.fsave_tag = pl.fsave_tag;
    const bool with_head = head != nullptr && head_done != nullptr && !spg_px_is_multi(pl.groups.n, N) && head->nin == (pl.cfg.cat_all ? 32 * (pl.R + 1) : 32) &&
                           head->C <= SPG_PX_HEAD_MAXC && spg_px_head_lds_bytes(*head) <= SPG_PX_HEAD_LDS;
    if (with_head) q.head = *head;
    int err = 0;
    if (spg_launch_ecc_persist_fwd(q, st, &err)) {
      if (with_head && err == 0) *head_done = true;
      return err;
    }
  }
  if (sc != nullptr) {      // per-iteration launches read a materialised descriptor matrix
    SPG_TRY(spg_gather_rows(sc->emb, 32, sc->slot_of_row, N, 32, sc->desc, 32, (void*)st));
    h0 = sc->desc;
  }
  if (pl.fsave_tag != nullptr) SPG_TRY(zero_async(pl.fsave_tag, sizeof(unsigned), st));      // per-iteration path: nothing was kept
  SPG_TRY(spg_launch_copy2d(h0, 32, pl.states, pl.ldS, N, 32, st));
  for (int r = 0; r < pl.R; ++r) {
    SpgEccStepFwd p; memset(&p, 0, sizeof(p));
    p.g = gr; p.W = pl.F.back().y; p.matrix = pl.cfg.matrix;
    p.hin = pl.states + (size_t)r * 32; p.hout = pl.states + (size_t)(r + 1) * 32; p.ld = pl.ldS;
    p.agg_save = pl.training ? pl.agg + (size_t)r * 32 : nullptr; p.ldagg = pl.ldS;
    p.do_gru = 1; p.gru = pl.gru; p.cell = pl.cfg.cell;
    if (pl.lstm) { p.cin = r > 0 ? pl.cells + (size_t)r * 32 : nullptr; p.cout = pl.cells + (size_t)(r + 1) * 32; }
    SPG_TRY(spg_launch_ecc_step_fwd(p, st));
  }
  if (pl.cfg.cat_all) SPG_TRY(spg_launch_copy2d(pl.states, pl.ldS, out, pl.ldS, N, (int)pl.ldS, st));
  else SPG_TRY(spg_launch_copy2d(pl.states + (size_t)pl.R * 32, pl.ldS, out, 32, N, 32, st));
  return 0;
}

// phase 0: the whole forward; phase 1: register the filter network's layers as riders (spg_gemm.h) and return -- they leave with
// the caller's next grouped launches; phase 2: the recurrent part only (the caller has drained the riders)
int spg_eccrnn_forward_phase(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* h0, const float* edgefeats,
                             const void* const* params, float* out, void* workspace, int training, int bn_update_times, void* stream,
                             int phase, const SpgEccScatter* sc, const SpgEccHead* head, bool* head_done) {
  SPG_CHECK_ARG(graph_ws && params && workspace && (phase == 1 || ((h0 || sc) && out)), "null pointer");
  SPG_CHECK_ARG(E == 0 || edgefeats != nullptr, "edgefeats");
  hipStream_t st = (hipStream_t)stream;
  Plan pl;
  SPG_TRY(make_plan(cfg, N, E, training, workspace, params, pl));
  pl.fold = fnet_fold(pl);
  // (as spg_pointnet.hip: with slot-synchronised BatchNorm the statistics MUST travel through the slots -- a rank that dropped to
  //  per-rank finalize statistics because its own edge count exceeds the slots' capacity, or because spg_tune key 10 is set, would
  //  issue a different number of slot all-reduces than its peers: a hang, or an unsynchronised model)
  if (pl.training && spg_slot_sync_active() && E > 0)
    for (const FLayer& l : pl.F)
      SPG_CHECK_ARG(!l.bn || pl.fold, "slot-synchronised BatchNorm needs the filter network's statistics slots (spg_tune key 10 off, edges within the slots' capacity on every rank)");
  if (E == 0 && pl.training && spg_slot_sync_active())
    for (const FLayer& l : pl.F) SPG_CHECK_ARG(!l.bn, "slot-synchronised BatchNorm: a rank with an edge-less batch cannot take part in the filter-network statistics");
  if (E == 0 && pl.training && spg_sync_bn_active())
    for (const FLayer& l : pl.F)      // the other ranks enter the layer's all-reduce: skipping it here would hang the job
      SPG_CHECK_ARG(!l.bn, "synchronised BatchNorm: a rank with an edge-less batch cannot take part in the filter-network statistics");
  if (phase != 2 && E > 0) {
    std::vector<SpgStage> stages;
    fnet_forward_stages(pl, edgefeats, bn_update_times, stages);
    if (phase == 1) {
      for (SpgStage& s : stages) spg_riders_push(std::move(s));
      return 0;
    }
    for (SpgStage& s : stages) {      // on their own: every layer is a launch (a group of one, + the slot clearing)
      SpgGroupScope grp(st);
      SPG_TRY(s(st));
      SPG_TRY(grp.flush());
    }
  }
  if (phase == 1) return 0;
  return eccrnn_recurrent_forward(pl, graph_ws, h0, out, st, sc, head, head_done);
}

extern "C" int spg_eccrnn_forward(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* h0,
                                  const float* edgefeats, const void* const* params, float* out, void* workspace,
                                  int training, int bn_update_times, void* stream) {
  return spg_eccrnn_forward_phase(cfg, N, E, graph_ws, h0, edgefeats, params, out, workspace, training, bn_update_times, stream, 0, nullptr, nullptr, nullptr);
}

extern "C" long spg_eccrnn_debug_offset(const spg_eccrnn_cfg* cfg, int N, int E, int training, int layer, int what) {
  Plan pl;
  char* fake = (char*)(uintptr_t)4096;
  if (make_plan(cfg, N, E, training, fake, nullptr, pl) != 0 || layer < 0 || layer >= (int)pl.F.size()) return -1;
  const FLayer& l = pl.F[layer];
  const void* p = what == 0 ? (const void*)l.y : what == 1 ? l.s : l.t;
  return p == nullptr ? -1 : (long)((const char*)p - fake);
}

extern "C" size_t spg_eccrnn_bwd_workspace_bytes(const spg_eccrnn_cfg* cfg, int N, int E) {
  Plan pl;
  if (make_plan(cfg, N, E, 1, nullptr, nullptr, pl) != 0) return 0;
  BwdScratch s;
  carve_bwd(pl, nullptr, s);
  return s.bytes;
}

// everything the backward's stages share (they may run after spg_eccrnn_backward_phase has returned: as riders)
namespace {
struct BwdCtx {
  Plan pl;
  BwdScratch s;
  SpgReduceQueue rq, rq2;
  SpgGraph gr;
  const float* edgefeats = nullptr;
  SpgOperand cur;
  int flip = 0;
  SpgBnFoldBwd pending;       // set by a data-gradient stage, finished by the next layer's launches (fold path)
};

// the tail of the backward: {cell parameter gradients + per-edge filter gradient}, then one stage per filter-network layer
// ({weight gradient, bias column sums, data gradient}: mutually independent), then the hand-over of the split partials
void eccrnn_backward_tail_stages(std::shared_ptr<BwdCtx> c, std::vector<SpgStage>& out, SpgStage extra_leaf) {
  // the recurrent cell's parameter gradients: three weight gradients + three bias column sums over all (node, iteration) rows.
  // A LEAF -- nothing in the chain below depends on them -- so they do not sit in front of the filter network's chain: they
  // leave with the stage of filter layer n-2 (their ~2000 small workgroups next to that layer's few)
  auto cell_grads = [c](hipStream_t st) -> int {
    Plan& pl = c->pl; BwdScratch& s = c->s;
    const int R = pl.R, GW = pl.GW, rows = pl.N * (R + 1);
    SpgWgradParams w; memset(&w, 0, sizeof(w));
    w.a = op_ident(s.dgi, GW); w.b = op_ident(s.xg, 32); w.M = rows; w.N = GW; w.K = 32;
    SPG_TRY(spg_queue_wgrad(c->rq2, w, pl.cell_grads[0], st));
    w.a = op_ident(s.dgh, GW); w.b = op_ident(pl.states, 32);
    SPG_TRY(spg_queue_wgrad(c->rq2, w, pl.cell_grads[1], st));
    // GRU: the biases are added behind the row normalisation; LSTM: in front of it
    SPG_TRY(spg_queue_colsum(c->rq2, pl.lstm ? s.dgi : s.dui, GW, rows, GW, pl.cell_grads[2], st));
    SPG_TRY(spg_queue_colsum(c->rq2, pl.lstm ? s.dgh : s.duh, GW, rows, GW, pl.cell_grads[3], st));
    if (pl.cfg.ingate) {
      w.a = op_ident(s.dpre, 32); w.b = op_ident(pl.states, 32); w.N = 32;
      // the input gate's bias gradient = column sums of dpre: as their own small job, not riding along with the weight
      // gradient -- the body that carries them needs more than 128 registers and would push the whole group (2700 small
      // workgroups) to the 2-workgroups-per-CU build of the grouped kernel (spg_gemm.hip)
      SPG_TRY(spg_queue_wgrad(c->rq2, w, pl.cell_grads[4], st));
      SPG_TRY(spg_queue_colsum(c->rq2, s.dpre, 32, rows, 32, pl.cell_grads[5], st));
    }
    return 0;
  };
  const int nF = (int)c->pl.F.size();
  const int leaf_with = c->pl.E > 0 && nF >= 2 ? nF - 2 : -1;      // the filter-layer stage the cell gradients travel with (-1: the first stage)
  // the caller's own leaf (spg_train_step: the classifier's parameter gradients -- a body of the 2-per-CU build of the grouped
  // kernel): with the LAST filter layer's stage, which needs that build anyway (in the first stage it pushed the per-edge filter
  // gradient's 1300 small workgroups off the 4-per-CU build: 14.5 -> 24.6 us; measured: +10 us whichever stage carries it)
  const int extra_with = c->pl.E > 0 && nF >= 1 ? nF - 1 : -1;
  out.push_back([c, cell_grads, leaf_with, extra_leaf, extra_with](hipStream_t st) -> int {
    Plan& pl = c->pl; BwdScratch& s = c->s;
    const int R = pl.R;
    const long ldS = pl.ldS;
    if (extra_leaf && extra_with < 0) SPG_TRY(extra_leaf(st));
    if (leaf_with < 0) SPG_TRY(cell_grads(st));
    if (pl.E == 0) {   // no edges: the filter network received no gradient
      for (FLayer& l : pl.F) {
        SPG_TRY(spg_group_zero(l.dW, (size_t)l.cin * l.cout, st));
        SPG_TRY(spg_group_zero(l.db, (size_t)l.cout, st));
        SPG_TRY(spg_group_zero(l.dgamma, (size_t)l.cout, st));
        SPG_TRY(spg_group_zero(l.dbeta, (size_t)l.cout, st));
      }
      return 0;
    }
    // per-edge filter gradients (sum over the iterations): the head of the filter network's chain
    SPG_TRY(spg_launch_ecc_edge_wgrad(c->gr, pl.cfg.matrix, pl.states, ldS, s.G, ldS, R, s.dWts, st));
    c->cur = op_ident(s.dWts, pl.nout);
    return 0;
  });
  for (int i = (int)c->pl.F.size() - 1; i >= 0 && c->pl.E > 0; --i) {
    out.push_back([c, i, cell_grads, leaf_with, extra_leaf, extra_with](hipStream_t st) -> int {
      Plan& pl = c->pl; BwdScratch& s = c->s;
      const int E = pl.E;
      FLayer& l = pl.F[i];
      if (i == leaf_with) SPG_TRY(cell_grads(st));
      if (extra_leaf && i == extra_with) SPG_TRY(extra_leaf(st));
      const SpgOperand cur = c->cur;
      const SpgBnFoldBwd fold_i = c->pending;
      memset(&c->pending, 0, sizeof(c->pending));
      SpgWgradParams w; memset(&w, 0, sizeof(w));
      w.a = cur; w.b = fnet_input(pl, i, c->edgefeats); w.M = E; w.N = l.cout; w.K = l.cin;
      w.fold = fold_i;
      const bool bias_rides = l.db != nullptr && !l.bn && cur.mode == SPG_PRO_IDENT;      // bias gradient = column sums of `cur`
      SPG_TRY(spg_queue_wgrad(c->rq, w, l.dW, st, bias_rides ? l.db : nullptr));
      if (l.db && !bias_rides) {
        if (l.bn) SPG_TRY(spg_group_zero(l.db, (size_t)l.cout, st));
        else SPG_TRY(spg_queue_colsum(c->rq, cur.X, cur.ld, E, l.cout, l.db, st));
      }
      if (i == 0) return 0;
      FLayer& prod = pl.F[i - 1];
      float* dzb[2] = {s.dzA, s.dzB};
      float* out = dzb[c->flip]; c->flip ^= 1;
      SpgGemmParams g; memset(&g, 0, sizeof(g));
      g.a = cur; g.W = l.W; g.ldw = l.cin; g.w_red = 1;
      g.M = E; g.N = l.cin; g.K = l.cout; g.rows_per_tile = SPG_FC_ROWS;
      g.epi = SPG_EPI_BWD; g.Y = out; g.ldy = l.cin; g.Yp = prod.y; g.ldyp = prod.cout;
      g.mask_relu = prod.relu ? 1 : 0; g.n_mask = prod.cout;
      g.fold_bwd = fold_i;      // (the weight gradient of this layer may run in the same grouped launch: both finish the constants)
      const bool fin = prod.bn && !pl.fold;      // finalize launch behind the data gradient: it must not wait in a group
      if (prod.bn) {
        g.ms = prod.s; g.mt = prod.t; g.mmean = prod.mean; g.mrstd = prod.rstd; g.stat = s.stat;
        if (pl.fold) { g.stat = nullptr; g.stat_slots = prod.slots_bwd; }
      }
      int nparts = 0;
      {
        SpgGroupBypass direct(fin);
        SPG_TRY(spg_launch_gemm(g, st, &nparts));
      }
      if (prod.bn) {
        if (fin) {
          SPG_TRY(spg_launch_bn_bwd_finalize(s.stat, nparts, l.cin, E, prod.cout, prod.s, prod.mean, prod.rstd,
                                             s.consts, prod.dgamma, prod.dbeta, nullptr, st));
        } else {
          SpgBnFoldBwd f; memset(&f, 0, sizeof(f));
          f.slots = prod.slots_bwd; f.C = prod.cout; f.count = (double)E; f.s = prod.s; f.mean = prod.mean; f.rstd = prod.rstd;
          if (spg_slot_sync_active()) f.grad_mul = 1.0 / (double)spg_slot_sync_world();
          f.consts = s.consts; f.dgamma = prod.dgamma; f.dbeta = prod.dbeta;
          c->pending = f;
        }
        c->cur = op_bnbwd(out, prod.y, prod.cout, s.consts, prod.cout);
      } else {
        c->cur = op_ident(out, l.cin);
      }
      return 0;
    });
  }
  // the split partials of all weight / bias gradients: handed to the NEXT batched reduction of this thread (spg_gemm.h:
  // spg_reduce_defer) -- PointNet's, when this chain rides next to its backward -- or summed by the caller's own flush
  out.push_back([c](hipStream_t st) -> int {
    for (int j = 0; j < c->rq2.njobs; ++j) spg_reduce_defer(c->rq2.jobs[j]);
    for (int j = 0; j < c->rq.njobs; ++j) spg_reduce_defer(c->rq.jobs[j]);
    c->rq.njobs = 0; c->rq2.njobs = 0;
    return 0;
  });
}
}  // namespace

// phase 0: the whole backward; phase 1: back-propagation through the recurrence (-> grad_h0), the tail (cell and filter-network
// parameter gradients) is registered as riders (spg_gemm.h): the caller lets them travel next to its next launches and MUST
// call spg_riders_drain + spg_flush_deferred_reduce before the gradients are consumed or the workspaces released
int spg_eccrnn_backward_phase(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* edgefeats,
                              const void* const* params, const float* grad_out, float* grad_h0, void* const* grads,
                              void* workspace, void* bwd_workspace, void* stream, int phase, const SpgEccScatter* sc,
                              SpgStage extra_leaf, const SpgEccHead* head) {
  SPG_CHECK_ARG(graph_ws && params && grad_out && grad_h0 && grads && workspace && bwd_workspace, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (phase == 0) spg_reduce_deferred_clear();      // (nothing may be left over from a call that failed half-way)
  auto c = std::make_shared<BwdCtx>();
  Plan& pl = c->pl;
  SPG_TRY(make_plan(cfg, N, E, 1, workspace, params, pl));
  pl.fold = fnet_fold(pl);
  if (spg_slot_sync_active() && E > 0)
    for (const FLayer& l : pl.F)
      SPG_CHECK_ARG(!l.bn || pl.fold, "slot-synchronised BatchNorm needs the filter network's statistics slots (spg_tune key 10 off, edges within the slots' capacity on every rank)");
  for (int i = 0; i < (int)pl.F.size(); ++i) {
    void* const* g = grads + 6 * i;
    pl.F[i].dW = (float*)g[0]; pl.F[i].db = (float*)g[1]; pl.F[i].dgamma = (float*)g[2]; pl.F[i].dbeta = (float*)g[3];
  }
  for (int k = 0; k < 6; ++k) pl.cell_grads[k] = (float*)grads[6 * pl.F.size() + k];
  BwdScratch& s = c->s;
  carve_bwd(pl, bwd_workspace, s);
  c->rq.arena = s.work; c->rq.arena_floats = s.work_floats;
  c->rq2.arena = s.work2; c->rq2.arena_floats = s.work2_floats;
  c->gr = spg_graph_view(graph_ws, N, E);
  c->edgefeats = edgefeats;
  memset(&c->cur, 0, sizeof(c->cur));
  memset(&c->pending, 0, sizeof(c->pending));
  SpgGraph& gr = c->gr;
  const int R = pl.R;
  const int GW = pl.GW;
  const long ldS = pl.ldS, ld96 = (long)(R + 1) * GW;
  // ---- back-propagation through the R iterations ----
  bool persistent = false;
  if (pl.px) {      // GRU: one dataflow-synchronised launch for all iterations (spg_ecc.hip)
    SpgEccPersistBwd q; memset(&q, 0, sizeof(q));
    q.groups = pl.groups;
    q.g = gr; q.W = pl.F.back().y; q.matrix = pl.cfg.matrix; q.R = R; q.cat_all = pl.cfg.cat_all;
    q.grad_out = grad_out; q.ldgo = pl.cfg.cat_all ? ldS : 32;
    q.states = pl.states; q.ldS = ldS; q.agg = pl.agg; q.G = s.G;
    q.dgi = s.dgi; q.dgh = s.dgh; q.dui = s.dui; q.duh = s.duh; q.ld96 = ld96;
    q.dpre = s.dpre; q.xg = s.xg; q.ld32 = ldS; q.gx = grad_h0; q.gru = pl.gru;
    if (sc != nullptr) { q.gx = sc->grad_emb; q.gx_rows = sc->slot_of_row; }      // the gradient gather written in place
    q.fsave = pl.fsave; q.fsave_tag = pl.fsave_tag;
    if (head != nullptr) q.head = *head;      // the head the forward launch ran: this launch sums its loss
    int err = 0;
    bool wgrad_done = false;
    persistent = spg_launch_ecc_persist_bwd(q, st, &err, head != nullptr ? &wgrad_done : nullptr);      // writes every row of [G .. xg] itself (slot R: zeros)
    if (err != 0) return err;
    if (wgrad_done) extra_leaf = SpgStage();      // the launch formed the classifier's parameter gradients on idle CUs
  }
  if (head != nullptr && !persistent)      // (the per-iteration fallback: the loss from the head's logits, as its own launch)
    SPG_TRY(spg_cross_entropy_fwd(head->logits, head->target, head->class_weight, N, head->C, head->ignore_index, head->reduction_mean,
                                  head->loss, head->lse, head->wsum, stream));
  if (!persistent) SPG_TRY(zero_async(bwd_workspace, s.zero_bytes, st));
  for (int r = R - 1; r >= 0 && !persistent; --r) {
    SpgEccStepBwd p; memset(&p, 0, sizeof(p));
    p.g = gr; p.W = pl.F.back().y; p.matrix = pl.cfg.matrix;
    if (pl.cfg.cat_all) { p.dcat = grad_out + (size_t)(r + 1) * 32; p.ldc = ldS; }
    else if (r == R - 1) { p.dcat = grad_out; p.ldc = 32; }
    p.Gnext = r < R - 1 ? s.G + (size_t)(r + 1) * 32 : nullptr; p.ldg = ldS;
    p.dhdir = s.dhdir; p.use_dhdir = r < R - 1;
    p.hin = pl.states + (size_t)r * 32; p.ld = ldS;
    p.agg = pl.agg + (size_t)r * 32; p.ldagg = ldS;
    p.Gcur = s.G + (size_t)r * 32;
    p.dgi = s.dgi + (size_t)r * GW; p.dgh = s.dgh + (size_t)r * GW; p.ld96 = ld96;
    if (!pl.lstm) { p.dui = s.dui + (size_t)r * 96; p.duh = s.duh + (size_t)r * 96; }
    p.dpre = s.dpre + (size_t)r * 32; p.xg = s.xg + (size_t)r * 32; p.ld32 = ldS;
    p.gru = pl.gru; p.cell = pl.cfg.cell;
    if (pl.lstm) { p.cin = r > 0 ? pl.cells + (size_t)r * 32 : nullptr; p.dcdir = s.dcdir; p.use_dcdir = r < R - 1; }
    SPG_TRY(spg_launch_ecc_step_bwd(p, st));
  }
  if (!persistent) {
    SpgEccStepBwd p; memset(&p, 0, sizeof(p));
    p.g = gr; p.W = pl.F.back().y; p.matrix = pl.cfg.matrix;
    if (pl.cfg.cat_all) { p.dcat = grad_out; p.ldc = ldS; }
    p.Gnext = s.G; p.ldg = ldS; p.dhdir = s.dhdir; p.use_dhdir = 1; p.gx = grad_h0; p.final_only = 1;
    p.gru = pl.gru; p.cell = pl.cfg.cell;
    SPG_TRY(spg_launch_ecc_step_bwd(p, st));
  }
  if (sc != nullptr && !persistent)       // per-iteration launches wrote grad_h0 [N, 32]: gather the embeddable rows
    SPG_TRY(spg_gather_rows(grad_h0, 32, sc->idx_valid, sc->B, 32, sc->grad_emb, 32, stream));
  // ---- the tail: nothing below depends on it, it depends on nothing but the recurrence's outputs ----
  std::vector<SpgStage> stages;
  eccrnn_backward_tail_stages(c, stages, std::move(extra_leaf));
  if (phase == 1) {
    for (SpgStage& sg : stages) spg_riders_push(std::move(sg));
    return 0;
  }
  for (SpgStage& sg : stages) {      // on their own: one grouped launch per stage
    SpgGroupScope grp(st);
    SPG_TRY(sg(st));
    SPG_TRY(grp.flush());
  }
  return spg_flush_deferred_reduce(st);     // ONE launch sums the split partials of all weight / bias gradients
}

extern "C" int spg_eccrnn_backward(const spg_eccrnn_cfg* cfg, int N, int E, const void* graph_ws, const float* edgefeats,
                                   const void* const* params, const float* grad_out, float* grad_h0, void* const* grads,
                                   void* workspace, void* bwd_workspace, void* stream) {
  return spg_eccrnn_backward_phase(cfg, N, E, graph_ws, edgefeats, params, grad_out, grad_h0, grads, workspace, bwd_workspace, stream, 0, nullptr, SpgStage(), nullptr);
}
