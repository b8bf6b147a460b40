// PointNet superpoint embedder on the fused row-GEMM kernels: host-side orchestration (kernel sequence,
// workspace layout) of the forward and backward passes.  Reference: learning/pointnet.py:16-133
// (STNkD, PointNet); CloudEmbedder's recompute-in-backward (pointnet.py:160-176) is replaced by keeping
// the raw layer outputs in HBM (288 GB per GPU), the BatchNorm running statistics are updated as many
// times as the reference's double forward would.
//
// Data layout in HBM
//   clouds         [B, nfeat, P]    as the loader produces them (channel-major per superpoint)
//   y_l            [B*P, C_l]       raw (pre-BatchNorm) output of 1x1-conv layer l, row = b*P + p
//   y_fc           [B, C]           raw output of an FC layer
//   BatchNorm(+ReLU) is applied by the CONSUMER while staging its tile into LDS (scale s, shift t per
//   channel), so normalised activations are never written.
//   The max-pool over the points is fused into the epilogue of the last conv (per-superpoint max AND
//   min of the raw output: after the batch statistics are known, sign(s) decides which one is the
//   max of the normalised value -- BatchNorm is monotone per channel).
#include "../../include/spg_hip.h"
#include "spg_ecc.h"
#include "spg_gemm.h"
#include "spg_narrow.h"
#include <limits.h>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
// Which Gram-slot arrays (one per workspace and segment: the device address is the key) the LAST training forward filled
// (spg_narrow.hip's one-pass first layers).  The backward of the first convolution takes its one-pass form only then: it used to
// re-evaluate the forward's predicates (spg_tune keys 17 / 18, the fold, the shapes) -- had they changed in between (A/B tooling
// toggling spg_tune), it would have read zeroed slots and written a wrong dW1 / dT without any error.
std::mutex g_gram_mu;
std::unordered_map<const void*, bool> g_gram_filled;
void gram_record(const void* slots, bool filled) { std::lock_guard<std::mutex> lock(g_gram_mu); g_gram_filled[slots] = filled; }
bool gram_recorded(const void* slots) {
  std::lock_guard<std::mutex> lock(g_gram_mu);
  auto it = g_gram_filled.find(slots);
  return it != g_gram_filled.end() && it->second;
}

struct Layer {
  int cin = 0, cout = 0;
  bool bn = false;
  bool conv = false;            // rows = points (B*P) instead of superpoints (B)
  const float *W = nullptr, *b = nullptr, *gamma = nullptr, *beta = nullptr;
  float *rm = nullptr, *rv = nullptr;
  float* y = nullptr;           // raw output (workspace or external)
  long ldy = 0;
  float* Wpack = nullptr;       // conv layers, inference: weights in MFMA operand order (spg_convstack.hip)
  void *Wb_f = nullptr, *Wb_t = nullptr;   // conv layers with whole reduction chunks: bf16 hi/lo copies of W for the opt-in bf16 MFMA
                                // modes (spg_launch_split_weights): forward orientation [2][cout][cin], transposed [2][cin][cout]
  float* Wpad = nullptr;        // FC layers whose input width is not a multiple of 4: zero-padded copy of W
  long ldw = 0;                 // leading dimension of the weight actually fed to the kernels
  float *mean = nullptr, *rstd = nullptr, *s = nullptr, *t = nullptr;   // BN batch constants
  unsigned long long* slots = nullptr;      // train mode: fixed-point statistics slots of this layer (SpgBnFold, spg_gemm.h)
  unsigned long long* slots_bwd = nullptr;  // the same for the backward sums (sum dz, sum dz * xhat)
  float *dW = nullptr, *db = nullptr, *dgamma = nullptr, *dbeta = nullptr;
};

struct Segment {           // convs (BN+ReLU each) -> max-pool (+concat) -> fcs (BN+ReLU each, last one plain)
  std::vector<int> convs, fcs;
  float* pooled = nullptr;  // [B, C_lastconv + nextra] selected raw max / min (+ concatenated global features)
  long ldpool = 0;
  int* aidx = nullptr;      // [B, C_lastconv]
  int nextra = 0;
  const float* extra = nullptr;
  unsigned long long* gram = nullptr;   // train mode: fixed-point slots of the Gram matrix of the segment's input (spg_narrow.h)
};

struct Plan {
  spg_pointnet_cfg cfg;
  int B = 0, P = 0;
  long M = 0;
  bool training = false;
  std::vector<Layer> L;
  Segment stn, main;
  bool has_stn = false;
  // shared scratch
  float *stat = nullptr, *stat_cnt = nullptr, *pmax = nullptr, *pmin = nullptr;
  double* fin = nullptr;    // scratch of the sliced BatchNorm finalize
  int *imax = nullptr, *imin = nullptr;
  unsigned long long* slots_all = nullptr;  // all layers' statistics slots, one region (cleared by one memset per forward)
  size_t slots_words = 0;
  bool fold = false;        // train mode: BatchNorm statistics are finished by the consuming GEMM instead of a finalize launch
  bool pads_done = false;   // the zero-padded weight copies of this forward have been issued
  size_t bytes = 0;
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

int num_layers(const spg_pointnet_cfg& c) {
  return (c.nfeat_stn > 0 ? c.n_stn_conv + c.n_stn_fc + 1 : 0) + c.n_conv + c.n_fc;
}

// builds the layer table and carves the forward workspace (base may be null: size query)
int make_plan(const spg_pointnet_cfg* cfg, int B, int training, void* ws, const void* const* params, float* emb,
              Plan& pl) {
  SPG_CHECK_ARG(cfg != nullptr && B > 0, "cfg / B");
  const spg_pointnet_cfg& c = *cfg;
  SPG_CHECK_ARG(c.npts >= 1 && c.npts <= 128, "npts must be in [1,128]");
  SPG_CHECK_ARG(c.n_conv >= 1 && c.n_fc >= 1 && c.n_conv <= SPG_MAX_LAYERS && c.n_fc <= SPG_MAX_LAYERS, "layer counts");
  SPG_CHECK_ARG(c.last_ac == 0, "last_ac=True is not supported");
  SPG_CHECK_ARG(c.nfeat_stn == 0 || (c.nfeat_stn >= 2 && c.nfeat_stn <= c.nfeat && c.n_stn_conv >= 1 && c.n_stn_fc >= 1),
                "STN configuration");
  SPG_CHECK_ARG(c.nfeat >= 1 && c.nfeat <= 32, "nfeat must be <= 32");
  pl.cfg = c; pl.B = B; pl.P = c.npts; pl.M = (long)B * c.npts; pl.training = training != 0;
  pl.has_stn = c.nfeat_stn > 0;
  pl.L.clear();
  auto add = [&](int cin, int cout, bool bn, bool conv) {
    Layer l; l.cin = cin; l.cout = cout; l.bn = bn; l.conv = conv;
    pl.L.push_back(l);
    return (int)pl.L.size() - 1;
  };
  pl.stn = Segment(); pl.main = Segment();
  if (pl.has_stn) {
    for (int i = 0; i < c.n_stn_conv; ++i) pl.stn.convs.push_back(add(i ? c.stn_conv[i - 1] : c.nfeat_stn, c.stn_conv[i], true, true));
    for (int i = 0; i < c.n_stn_fc; ++i)
      pl.stn.fcs.push_back(add(i ? c.stn_fc[i - 1] : c.stn_conv[c.n_stn_conv - 1], c.stn_fc[i], true, false));
    pl.stn.fcs.push_back(add(c.stn_fc[c.n_stn_fc - 1], 4, false, false));   // proj
  }
  for (int i = 0; i < c.n_conv; ++i) pl.main.convs.push_back(add(i ? c.conv[i - 1] : c.nfeat, c.conv[i], true, true));
  for (int i = 0; i < c.n_fc; ++i)
    pl.main.fcs.push_back(add(i ? c.fc[i - 1] : c.conv[c.n_conv - 1] + c.nfeat_global, c.fc[i], i < c.n_fc - 1, false));
  pl.main.nextra = c.nfeat_global;

  Carver cv(ws);
  int cmax = 4;
  for (size_t i = 0; i < pl.L.size(); ++i) {
    Layer& l = pl.L[i];
    if (params) {
      const void* const* g = params + 6 * i;
      l.W = (const float*)g[0]; l.b = (const float*)g[1]; l.gamma = (const float*)g[2]; l.beta = (const float*)g[3];
      l.rm = (float*)g[4]; l.rv = (float*)g[5];
      SPG_CHECK_ARG(l.W != nullptr, "missing layer weight");
      SPG_CHECK_ARG(!l.bn || (l.rm != nullptr && l.rv != nullptr), "missing BatchNorm running statistics");
    }
    cmax = l.cout > cmax ? l.cout : cmax;
    if (l.bn) {
      l.mean = cv.take<float>(l.cout); l.rstd = cv.take<float>(l.cout);
      l.s = cv.take<float>(l.cout); l.t = cv.take<float>(l.cout);
    }
    l.ldw = l.cin;
    if (!l.conv && (l.cin & 3) != 0) {     // e.g. 256 pooled channels + 1 diameter = 257
      l.ldw = (l.cin + 3) & ~3;
      l.Wpad = cv.take<float>((size_t)l.cout * l.ldw);
    }
    // training, first convolution of the main segment behind an STN: its data gradient wrt the transformed xy reads the weight
    // untransposed [cout][cin]; with cin no multiple of 4 (14, 11) a zero-padded copy keeps it on the vector staging path
    if (l.conv && pl.training && pl.has_stn && (int)i == (pl.has_stn ? c.n_stn_conv + c.n_stn_fc + 1 : 0) && (l.cin & 3) != 0) {
      l.ldw = (l.cin + 3) & ~3;
      l.Wpad = cv.take<float>((size_t)l.cout * l.ldw);
    }
    if (l.conv && !pl.training) l.Wpack = cv.take<float>(spg_conv_stack_packed_floats(l.cin, l.cout));
    if (l.conv && pl.training && l.cin % SPG_KC == 0) {      // layers the full-tile GEMM path can take (whole reduction chunks)
      l.Wb_f = cv.take<char>(spg_split_bytes(l.cout, l.cin));
      l.Wb_t = cv.take<char>(spg_split_bytes(l.cin, l.cout));
    }
  }
  if (pl.training) {
    size_t words = 0;
    for (const Layer& l : pl.L) if (l.bn) words += 2 * spg_fold_slot_words(l.cout);
    // + per segment the slots of the input's Gram matrix (the first two convolutions as one pass, spg_narrow.hip): inside the
    // same region, so they are cleared with the layers' slots
    const size_t gw_stn = pl.has_stn ? spg_gram_slot_words(c.nfeat_stn) : 0, gw_main = spg_gram_slot_words(c.nfeat);
    words += gw_stn + gw_main;
    pl.slots_all = cv.take<unsigned long long>(words); pl.slots_words = words;
    size_t off = 0;
    for (Layer& l : pl.L)
      if (l.bn) {
        l.slots = pl.slots_all ? pl.slots_all + off : nullptr; off += spg_fold_slot_words(l.cout);
        l.slots_bwd = pl.slots_all ? pl.slots_all + off : nullptr; off += spg_fold_slot_words(l.cout);
      }
    if (pl.has_stn) { pl.stn.gram = pl.slots_all ? pl.slots_all + off : nullptr; off += gw_stn; }
    pl.main.gram = pl.slots_all ? pl.slots_all + off : nullptr; off += gw_main;
  }
  auto carve_segment = [&](Segment& sg, float* final_out) {
    for (size_t k = 0; k < sg.convs.size(); ++k) {
      Layer& l = pl.L[sg.convs[k]];
      const bool last = k + 1 == sg.convs.size();
      l.ldy = l.cout;
      // the pooled layer's dense output is only needed by the backward pass
      l.y = (!last || pl.training) ? cv.take<float>((size_t)pl.M * l.cout) : nullptr;
    }
    const Layer& lc = pl.L[sg.convs.back()];
    sg.ldpool = (lc.cout + sg.nextra + 3) & ~3;      // padded: rows readable as aligned float4
    sg.pooled = cv.take<float>((size_t)B * sg.ldpool);
    sg.aidx = cv.take<int>((size_t)B * sg.ldpool);
    for (size_t k = 0; k < sg.fcs.size(); ++k) {
      Layer& l = pl.L[sg.fcs[k]];
      const bool last = k + 1 == sg.fcs.size();
      l.ldy = l.cout;
      l.y = (last && final_out) ? final_out : cv.take<float>((size_t)B * l.cout);
    }
  };
  if (pl.has_stn) carve_segment(pl.stn, nullptr);
  carve_segment(pl.main, emb);
  pl.stat = cv.take<float>((size_t)B * 4 * 2 * cmax);      // up to 4 per-wave partials per tile
  pl.stat_cnt = cv.take<float>((size_t)B * 4 + 64);
  pl.fin = cv.take<double>(spg_bn_finalize_scratch_doubles(cmax));
  pl.pmax = cv.take<float>((size_t)B * 4 * cmax); pl.pmin = cv.take<float>((size_t)B * 4 * cmax);
  pl.imax = cv.take<int>((size_t)B * 4 * cmax); pl.imin = cv.take<int>((size_t)B * 4 * cmax);
  pl.bytes = cv.off + 256;
  return 0;
}

SpgOperand op_affine(const Layer& prod, const float* X, long ld, int n_affine) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_AFFINE; o.X = X; o.ld = ld; o.c0 = prod.s; o.c1 = prod.t; o.relu = 1; o.n_affine = n_affine;
  return o;
}
SpgOperand op_cloud(const Plan& pl, const float* clouds, const float* stnT) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_CLOUD; o.X = clouds; o.Ctot = pl.cfg.nfeat; o.P = pl.P; o.stnT = stnT;
  return o;
}
SpgOperand op_ident(const float* X, long ld) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_IDENT; o.X = X; o.ld = ld;
  return o;
}

// the operand through which layer `k` of the segment reads its input in the forward pass
SpgOperand input_operand(const Plan& pl, const Segment& sg, bool is_fc, size_t k, const float* clouds, const float* stnT) {
  if (!is_fc) {
    if (k == 0) return op_cloud(pl, clouds, stnT);
    const Layer& p = pl.L[sg.convs[k - 1]];
    return op_affine(p, p.y, p.ldy, p.cout);
  }
  if (k == 0) {
    const Layer& p = pl.L[sg.convs.back()];
    return op_affine(p, sg.pooled, sg.ldpool, p.cout);
  }
  const Layer& p = pl.L[sg.fcs[k - 1]];
  return op_affine(p, p.y, p.ldy, p.cout);
}

// the statistics of `prod` (rows = `count`) finished in the prologue of the launch that consumes its output
SpgBnFold fold_of(const Plan& pl, const Layer& prod, long count, int update_times) {
  SpgBnFold f; memset(&f, 0, sizeof(f));
  if (!pl.fold || !prod.bn) return f;
  f.slots = prod.slots; f.C = prod.cout; f.update_times = update_times; f.momentum = pl.cfg.bn_momentum; f.eps = pl.cfg.bn_eps;
  f.count = (double)count; f.gamma = prod.gamma; f.beta = prod.beta; f.rm = prod.rm; f.rv = prod.rv;
  f.mean = prod.mean; f.rstd = prod.rstd; f.s = prod.s; f.t = prod.t;
  return f;
}

int bn_stats(const Plan& pl, Layer& l, int nparts, long M, int update_times, hipStream_t st) {
  if (pl.training && pl.fold) return 0;      // finished by the consumer (fold_of)
  if (pl.training)
    return spg_launch_bn_finalize(pl.stat, pl.stat_cnt, nparts, M, l.cout, l.gamma, l.beta, l.rm, l.rv,
                                  pl.cfg.bn_momentum, pl.cfg.bn_eps, update_times, l.mean, l.rstd, l.s, l.t, pl.fin, st);
  return 0;     // eval mode: all layers were handled by one spg_launch_bn_eval_batch at the start of the forward
}

// inference: the whole convolution stack + max-pool of a segment in one kernel (spg_convstack.hip), when its shape allows
bool fused_eval_convs(Plan& pl, Segment& sg, const float* clouds, const float* stnT, SpgConvStackParams& cp) {
  if (pl.training || sg.convs.size() > SPG_CONVSTACK_MAX_LAYERS) return false;
  memset(&cp, 0, sizeof(cp));
  cp.clouds = clouds; cp.stnT = stnT; cp.B = pl.B; cp.P = pl.P; cp.Ctot = pl.cfg.nfeat; cp.nlayers = (int)sg.convs.size();
  for (size_t k = 0; k < sg.convs.size(); ++k) {
    const Layer& l = pl.L[sg.convs[k]];
    if (!l.bn) return false;
    cp.cin[k] = l.cin; cp.cout[k] = l.cout; cp.W[k] = l.W; cp.bias[k] = l.b; cp.s[k] = l.s; cp.t[k] = l.t;
    cp.Wp[k] = reinterpret_cast<f32x4*>(l.Wpack);
  }
  cp.pmax = pl.pmax; cp.pmin = pl.pmin;
  return spg_conv_stack_eval_supported(cp);
}

int forward_segment(Plan& pl, Segment& sg, const float* clouds, const float* stnT, int update_times, hipStream_t st) {
  SpgConvStackParams cp;
  const bool fused = fused_eval_convs(pl, sg, clouds, stnT, cp);
  if (fused) {
    const Layer& ll = pl.L[sg.convs.back()];
    SPG_TRY(spg_launch_conv_stack_eval(cp, st));
    SPG_TRY(spg_launch_pool_select_parts(pl.pmax, pl.pmin, nullptr, nullptr, ll.s, pl.B, ll.cout, 4, sg.extra, sg.nextra,
                                         sg.pooled, sg.ldpool, nullptr, st));
  }
  size_t kfirst = 0;
  if (pl.training && sg.gram != nullptr) gram_record(sg.gram, false);      // (set below when this forward fills the Gram slots)
  // train mode: the first two convolutions (cloud -> 64 -> 64) as ONE pass over the points -- the first layer's batch statistics
  // from the Gram matrix of the input, its raw output written once and never read back (spg_narrow.hip)
  if (!fused && pl.training && pl.fold && sg.gram != nullptr && sg.convs.size() > 2) {
    Layer& l0 = pl.L[sg.convs[0]];
    Layer& l1 = pl.L[sg.convs[1]];
    if (l0.bn && l1.bn && spg_narrow_pair_supported(l0.cin, l0.cout, l1.cout, pl.P, pl.M)) {
      SpgGramParams gp; memset(&gp, 0, sizeof(gp));
      gp.clouds = clouds; gp.stnT = stnT; gp.B = pl.B; gp.P = pl.P; gp.Ctot = pl.cfg.nfeat; gp.nfeat = l0.cin; gp.gram = sg.gram;
      SPG_TRY(spg_launch_cloud_gram(gp, st));
      SPG_TRY(spg_slot_sync_after(sg.gram, spg_gram_slot_words(l0.cin), st, false));      // (slot-synchronised BatchNorm)
      SpgNarrowPairParams np; memset(&np, 0, sizeof(np));
      np.clouds = clouds; np.stnT = stnT; np.P = pl.P; np.Ctot = pl.cfg.nfeat; np.nfeat = l0.cin; np.nblk = (int)(pl.M / 32);
      np.count = (double)pl.M;
      np.W1 = l0.W; np.b1 = l0.b; np.y1 = l0.y; np.gram = sg.gram; np.gamma1 = l0.gamma; np.beta1 = l0.beta; np.rm1 = l0.rm; np.rv1 = l0.rv;
      np.mean1 = l0.mean; np.rstd1 = l0.rstd; np.s1 = l0.s; np.t1 = l0.t;
      np.update_times = update_times; np.momentum = pl.cfg.bn_momentum; np.eps = pl.cfg.bn_eps;
      np.W2 = l1.W; np.b2 = l1.b; np.y2 = l1.y; np.slots2 = l1.slots;
      SPG_TRY(spg_launch_narrow_pair_fwd(np, st));
      SPG_TRY(spg_slot_sync_after(l1.slots, spg_fold_slot_words(l1.cout), st, false));
      gram_record(sg.gram, true);
      kfirst = 2;
    }
  }
  for (size_t k = kfirst; !fused && k < sg.convs.size(); ++k) {
    Layer& l = pl.L[sg.convs[k]];
    const bool last = k + 1 == sg.convs.size();
    SpgGemmParams g; memset(&g, 0, sizeof(g));
    g.a = input_operand(pl, sg, false, k, clouds, stnT);
    if (k > 0) g.fold = fold_of(pl, pl.L[sg.convs[k - 1]], pl.M, update_times);
    g.W = l.W; g.ldw = l.cin; g.bias = l.b; g.M = (int)pl.M; g.N = l.cout; g.K = l.cin;
    g.rows_per_tile = pl.P; g.epi = SPG_EPI_FWD; g.Y = l.y; g.ldy = l.ldy;
    if (l.Wb_f != nullptr) { g.Wb = l.Wb_f; g.ldwb = spg_split_ld(l.cin); g.wb_part_bytes = (long)l.cout * g.ldwb * 2; }
    if (last && !pl.training) g.Y = nullptr;     // inference: only the pooled values of the last conv are consumed
    g.stat = pl.training ? pl.stat : nullptr; g.stat_cnt = pl.stat_cnt;
    if (pl.fold) { g.stat = nullptr; g.stat_slots = l.slots; }
    int nparts = 0;
    if (last) {      // max-pool fused into the epilogue: the sign of the BatchNorm scale is the sign of gamma
      g.pool_out = sg.pooled; g.pool_idx = sg.aidx; g.pool_ld = sg.ldpool; g.pool_sign = l.gamma;
      g.pool_extra = sg.extra; g.pool_nextra = sg.nextra;
    }
    SPG_TRY(spg_launch_gemm(g, st, &nparts));
    SPG_TRY(bn_stats(pl, l, nparts, pl.M, update_times, st));
  }
  // every FC layer is a few-row launch: issued through a group scope (a group of one, or of two with the weight padding below),
  // so that a rider chain of the caller -- the filter network's forward -- can leave with it (spg_gemm.h)
  SpgGroupScope grp(st);
  for (size_t k = 0; k < sg.fcs.size(); ++k) {
    Layer& l = pl.L[sg.fcs[k]];
    SpgGemmParams g; memset(&g, 0, sizeof(g));
    g.a = input_operand(pl, sg, true, k, clouds, stnT);
    // the producer's statistics: the last convolution (over all points) for the first fc layer, else the previous fc layer
    g.fold = k == 0 ? fold_of(pl, pl.L[sg.convs.back()], pl.M, update_times) : fold_of(pl, pl.L[sg.fcs[k - 1]], pl.B, update_times);
    // zero-padded weight copies (input widths that are no multiples of 4: 256 pooled channels + the diameter): they depend on
    // the weights only -- all of them leave with the FIRST few-row launch of the forward that does not need one itself
    if (!pl.pads_done) {
      bool mine = false;
      for (int idx : sg.fcs) mine = mine || pl.L[idx].Wpad != nullptr;
      if (mine && k == 0) {      // this segment's own layers need them: a launch of their own, in front
        for (Layer& q : pl.L) if (q.Wpad) SPG_TRY(spg_launch_pad_rows(q.W, q.cin, q.Wpad, q.ldw, q.cout, q.cin, st));
        SPG_TRY(grp.flush());
      } else if (!mine) {
        for (Layer& q : pl.L) if (q.Wpad) SPG_TRY(spg_launch_pad_rows(q.W, q.cin, q.Wpad, q.ldw, q.cout, q.cin, st));
      }
      pl.pads_done = true;
    }
    g.W = l.Wpad ? l.Wpad : l.W; g.ldw = l.ldw; g.bias = l.b; g.M = pl.B; g.N = l.cout; g.K = l.cin;
    g.rows_per_tile = SPG_FC_ROWS; g.epi = SPG_EPI_FWD; g.Y = l.y; g.ldy = l.ldy;
    g.stat = (pl.training && l.bn) ? pl.stat : nullptr; g.stat_cnt = pl.stat_cnt;
    if (pl.fold && l.bn) { g.stat = nullptr; g.stat_slots = l.slots; }
    int nparts = 0;
    {
      SpgGroupBypass direct(l.bn && pl.training && !pl.fold);      // a finalize launch follows: the GEMM must not wait in the group
      SPG_TRY(spg_launch_gemm(g, st, &nparts));
    }
    SPG_TRY(grp.flush());
    if (l.bn) SPG_TRY(bn_stats(pl, l, nparts, pl.B, update_times, st));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
struct BwdScratch {
  float *dzA = nullptr, *dzB = nullptr;     // [M, cmax_conv] ping-pong for the dense conv gradients
  float *fzA = nullptr, *fzB = nullptr;     // [B, cmax_fc]
  float* consts = nullptr;                  // [4][cmax]
  float* consts_leaf[2] = {nullptr, nullptr};   // the BatchNorm-backward constants of the layers whose weight gradient leaves later as
                                            // leaves (spg_gemm.h): `consts` is rewritten by every layer in between
  float *fzC = nullptr, *fzD = nullptr;     // the STN head's own ping-pong: the main segment's pooled gradient (fzA / fzB) is still
                                            // read by the pooled convolution's weight-gradient leaves while the STN head runs
  double* fin = nullptr;                    // scratch of the sliced finalize
  float* work = nullptr;                    // reduction arena: split partials of all weight / bias gradients
  size_t work_floats = 0;
  float* stat = nullptr;                    // [ntile][2][cmax]
  float* dxy = nullptr;                     // [M, 2]
  float* dT = nullptr;                      // [B, 4]
  float* grad_global = nullptr;             // optional output [B, nextra]: gradient wrt the concatenated global features
  size_t bytes = 0;
};

void carve_bwd(const Plan& pl, void* ws, BwdScratch& s) {
  Carver cv(ws);
  int cconv = 4, cfc = 4, cmax = 4;
  size_t wmax = 16, workmax = 16;
  for (const Layer& l : pl.L) {
    if (l.conv) { cconv = l.cin > cconv ? l.cin : cconv; }
    else { cfc = l.cin + 4 > cfc ? l.cin + 4 : cfc; }
    cmax = l.cout > cmax ? l.cout : cmax; cmax = l.cin > cmax ? l.cin : cmax;
    wmax = (size_t)l.cin * l.cout > wmax ? (size_t)l.cin * l.cout : wmax;
    // every layer gets its own slice of the reduction arena (partials stay alive until the single batched reduce)
    workmax += ((spg_wgrad_workspace_floats(l.conv ? pl.M : pl.B, l.cout, l.cin) + 63) & ~(size_t)63) + 64 * (size_t)l.cout + 128 +
               ((spg_wgrad_colsum_floats(l.conv ? pl.M : pl.B, l.cout, l.cin) + 63) & ~(size_t)63);
    // (the one-pass backward of a segment's first convolution writes one partial per wave, spg_narrow.h)
    if (l.conv && l.cin <= SPG_GRAM_MAXF) workmax += ((size_t)spg_first_conv_bwd_partials(pl.B) * l.cout * l.cin + 63) & ~(size_t)63;
  }
  s.dzA = cv.take<float>((size_t)pl.M * cconv); s.dzB = cv.take<float>((size_t)pl.M * cconv);
  s.fzA = cv.take<float>((size_t)pl.B * cfc); s.fzB = cv.take<float>((size_t)pl.B * cfc);
  s.consts = cv.take<float>((size_t)4 * cmax);
  s.consts_leaf[0] = cv.take<float>((size_t)4 * cmax); s.consts_leaf[1] = cv.take<float>((size_t)4 * cmax);
  s.fzC = cv.take<float>((size_t)pl.B * cfc); s.fzD = cv.take<float>((size_t)pl.B * cfc);
  s.fin = cv.take<double>(spg_bn_finalize_scratch_doubles(cmax));
  s.work = cv.take<float>(workmax); s.work_floats = workmax;
  s.stat = cv.take<float>((size_t)pl.B * 4 * 2 * cmax);
  s.dxy = cv.take<float>((size_t)pl.M * 2);
  s.dT = cv.take<float>((size_t)pl.B * 4);
  s.bytes = cv.off + 256;
}

SpgOperand op_bnbwd(const float* dz, const float* y, long ld, const float* consts, int C) {
  SpgOperand o; memset(&o, 0, sizeof(o));
  o.mode = SPG_PRO_BNBWD; o.X = dz; o.X2 = y; o.ld = ld;
  o.c0 = consts; o.c1 = consts + C; o.c2 = consts + 2 * C; o.c3 = consts + 3 * C;
  return o;
}

int zero_async(float* p, size_t n, hipStream_t st) {
  if (p == nullptr || n == 0) return 0;
  hipError_t e = hipMemsetAsync(p, 0, n * sizeof(float), st);
  if (e != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// Backward of one segment.  `cur` is the gradient wrt the raw output of the segment's last fc layer
// (IDENT operand).  If want_dxy, the gradient wrt the first two input channels of conv 0 is left in s.dxy.
// the BatchNorm-backward sums of `prod` (rows = count), to be finished by the next weight-gradient launch (spg_gemm.h)
SpgBnFoldBwd fold_bwd_of(const Plan& pl, const Layer& prod, long count, float* consts) {
  SpgBnFoldBwd f; memset(&f, 0, sizeof(f));
  f.slots = prod.slots_bwd; f.C = prod.cout; f.count = (double)count; f.s = prod.s; f.mean = prod.mean; f.rstd = prod.rstd;
  f.consts = consts; f.dgamma = prod.dgamma; f.dbeta = prod.dbeta;
  if (spg_slot_sync_active()) f.grad_mul = 1.0 / (double)spg_slot_sync_world();
  return f;
}

// leaves: 1 = this segment's convolutions hand the weight gradients that are not fused with their data gradient (the pooled
// layer's, the first layer's) to spg_queue_wgrad_leaf (a later segment's head takes them along); 2 = this segment's head takes
// pending leaves along (and uses the second pair of head buffers)
// dT_out / dT_done: with want_dxy, the one-pass backward of the first convolution (spg_narrow.h) writes the gradient of the 2 x 2
// transforms [B, 4] itself and sets *dT_done; otherwise s.dxy holds the gradient wrt the transformed xy for spg_launch_stn_dT
int backward_segment(Plan& pl, Segment& sg, BwdScratch& s, SpgReduceQueue& rq, SpgOperand cur, const float* clouds,
                     const float* stnT, bool want_dxy, hipStream_t st, bool ride_reduce = false, int leaves = 0,
                     float* dT_out = nullptr, bool* dT_done = nullptr) {
  const int B = pl.B;
  SpgBnFoldBwd pending; memset(&pending, 0, sizeof(pending));     // set by a data-gradient launch, consumed by the next weight gradient
  // ---- fc head ----
  // A layer's weight gradient and its data gradient depend on the same inputs and not on each other: both few-row launches
  // leave as ONE grouped launch (spg_gemm.h) -- three launches per head instead of six.  The data gradient then finishes the
  // BatchNorm-backward constants of its operand itself (fold_bwd), since the weight gradient is no longer ordered before it.
  float* fz[2] = {leaves == 2 ? s.fzC : s.fzA, leaves == 2 ? s.fzD : s.fzB};
  int flip = 0;
  {
  SpgGroupScope grp(st);
  // the split partials the launches so far have queued (the other segment's weight gradients: ~100 MB, 2/3 of the step's final
  // reduction) are summed as jobs of this segment's grouped launches -- a slice of ~35 MB each -- in the shadow of its
  // latency-bound FC layers
  for (int k = (int)sg.fcs.size() - 1; k >= 0; --k) {
    if (ride_reduce) SPG_TRY(spg_reduce_ride(rq, st));      // (a slice per group; before this layer queues its own partials)
    Layer& l = pl.L[sg.fcs[k]];
    SpgWgradParams w; memset(&w, 0, sizeof(w));
    w.a = cur; w.b = input_operand(pl, sg, true, k, clouds, stnT); w.M = B; w.N = l.cout; w.K = l.cin;
    w.fold = pending;
    const SpgBnFoldBwd fold_k = pending;
    memset(&pending, 0, sizeof(pending));
    // a bias without BatchNorm behind it: its gradient (column sums of `cur`, an IDENT operand here) rides along with the
    // weight gradient; a bias in front of train-mode BatchNorm has zero gradient
    const bool bias_rides = l.db != nullptr && !l.bn && cur.mode == SPG_PRO_IDENT;
    SPG_TRY(spg_queue_wgrad(rq, w, l.dW, st, bias_rides ? l.db : nullptr));
    if (l.db && !bias_rides) {
      if (l.bn) SPG_TRY(spg_group_zero(l.db, l.cout, st));
      else SPG_TRY(spg_queue_colsum(rq, cur.X, cur.ld, B, l.cout, l.db, st));
    }
    // data gradient -> producer of this layer's input
    const bool first = k == 0;
    Layer& prod = first ? pl.L[sg.convs.back()] : pl.L[sg.fcs[k - 1]];
    float* out = fz[flip]; flip ^= 1;
    SpgGemmParams g; memset(&g, 0, sizeof(g));
    g.a = cur; g.W = l.Wpad ? l.Wpad : l.W; g.ldw = l.ldw; g.w_red = 1;   // dz_prev = dy @ W, W read untransposed
    g.M = B; g.N = l.cin; g.K = l.cout; g.rows_per_tile = SPG_FC_ROWS;
    g.epi = SPG_EPI_BWD; g.Y = out; g.ldy = first ? sg.ldpool : l.cin;
    g.Yp = first ? sg.pooled : prod.y; g.ldyp = first ? sg.ldpool : prod.ldy;
    g.ms = prod.s; g.mt = prod.t; g.mask_relu = 1; g.n_mask = prod.cout;
    g.mmean = prod.mean; g.mrstd = prod.rstd; g.stat = s.stat;
    if (pl.fold) { g.stat = nullptr; g.stat_slots = prod.slots_bwd; g.stat_rows = first ? pl.M : (long)B; }
    if (grp.active()) g.fold_bwd = fold_k;
    int nparts = 0;
    SPG_TRY(spg_launch_gemm(g, st, &nparts));
    if (leaves == 2) SPG_TRY(spg_leaf_ride(st, k + 1));      // a share of the pending leaves (this head has k + 1 launches left)
    SPG_TRY(grp.flush());      // {weight gradient, bias column sums, data gradient} of this layer
    // the statistics cover the producer's channels only (N = l.cin may be larger by nextra for the pooled input)
    const int C = prod.cout;
    if (first && s.grad_global != nullptr && sg.nextra > 0)     // columns >= C pass through: gradient wrt the global features
      SPG_TRY(spg_launch_copy2d(out + C, sg.ldpool, s.grad_global, sg.nextra, B, sg.nextra, st));
    if (pl.fold) pending = fold_bwd_of(pl, prod, first ? pl.M : (long)B, s.consts);      // finished by the next weight gradient
    else
    SPG_TRY(spg_launch_bn_bwd_finalize(s.stat, nparts, l.cin, first ? pl.M : (long)B, C, prod.s, prod.mean,
                                       prod.rstd, s.consts, prod.dgamma, prod.dbeta, s.fin, st));
    if (!first) {
      cur = op_bnbwd(out, prod.y, prod.ldy, s.consts, C);
    } else {
      memset(&cur, 0, sizeof(cur));
      cur.mode = SPG_PRO_POOLBWD; cur.X = out; cur.ldg = (int)sg.ldpool; cur.aidx = sg.aidx; cur.P = pl.P;
      cur.X2 = prod.y; cur.ld = prod.ldy;
      cur.c0 = s.consts; cur.c1 = s.consts + C; cur.c2 = s.consts + 2 * C; cur.c3 = s.consts + 3 * C;
    }
  }
  }      // (the group scope ends with the head: the convolutions' launches are ordered, never grouped)
  // ---- conv stack ----
  float* dz[2] = {s.dzA, s.dzB};
  flip = 0;
  for (int k = (int)sg.convs.size() - 1; k >= 0; --k) {
    Layer& l = pl.L[sg.convs[k]];
    if (k > 0 && pl.fold) {      // (pending: this layer's sums, or none when a finalize launch already made the constants)
      // 64-input-channel layers: data gradient + weight gradient from ONE pass over dz (spg_gemm.h: spg_queue_bwdpair)
      Layer& prod = pl.L[sg.convs[k - 1]];
      float* out = dz[flip];
      SpgGemmParams g; memset(&g, 0, sizeof(g));
      g.a = cur; g.W = l.W; g.ldw = l.cin; g.w_red = 1;
      g.M = (int)pl.M; g.N = l.cin; g.K = l.cout; g.rows_per_tile = pl.P;
      g.epi = SPG_EPI_BWD; g.Y = out; g.ldy = l.cin; g.Yp = prod.y; g.ldyp = prod.ldy;
      g.ms = prod.s; g.mt = prod.t; g.mask_relu = 1; g.n_mask = prod.cout;
      g.mmean = prod.mean; g.mrstd = prod.rstd; g.stat_slots = prod.slots_bwd; g.fold_bwd = pending;
      const SpgOperand xin = input_operand(pl, sg, false, k, clouds, stnT);
      if (spg_bwdpair_supported(g, xin)) {
        SPG_TRY(spg_queue_bwdpair(rq, g, xin, l.dW, st));
        if (l.db) SPG_TRY(zero_async(l.db, l.cout, st));
        flip ^= 1;
        pending = fold_bwd_of(pl, prod, pl.M, s.consts);      // (the launch consumed this layer's; these are the producer's)
        cur = op_bnbwd(out, prod.y, prod.ldy, s.consts, prod.cout);
        continue;
      }
    }
    // The FIRST convolution behind a forward that left the Gram matrix of the input in the slots (spg_narrow.hip): weight gradient
    // and -- main segment behind an STN -- the gradient of the 2 x 2 transforms from ONE pass over the incoming gradient and the
    // cloud; the layer's raw output is not read (it is linear in the cloud: its part collapses onto the Gram matrix)
    if (k == 0 && pl.fold && sg.gram != nullptr && gram_recorded(sg.gram) && sg.convs.size() > 2 && pending.slots != nullptr && cur.mode == SPG_PRO_BNBWD &&
        cur.ld == l.cout && cur.c0 == s.consts && (!want_dxy || dT_out != nullptr) &&
        spg_narrow_pair_supported(l.cin, l.cout, pl.L[sg.convs[1]].cout, pl.P, pl.M) && spg_first_conv_bwd_supported(l.cin, l.cout, pl.P, pl.M)) {
      SpgFirstConvBwdParams fp; memset(&fp, 0, sizeof(fp));
      fp.clouds = clouds; fp.stnT = stnT; fp.B = pl.B; fp.P = pl.P; fp.Ctot = pl.cfg.nfeat; fp.nfeat = l.cin;
      fp.g = cur.X; fp.W1 = l.W; fp.gram = sg.gram; fp.fold = pending; memset(&pending, 0, sizeof(pending));
      fp.dT = want_dxy ? dT_out : nullptr;
      SPG_TRY(spg_queue_partials(rq, spg_first_conv_bwd_partials(pl.B), l.cout * l.cin, l.dW, &fp.partial, st));
      SPG_TRY(spg_launch_first_conv_bwd(fp, st));
      if (l.db) SPG_TRY(zero_async(l.db, l.cout, st));
      if (want_dxy && dT_done != nullptr) *dT_done = true;
      continue;
    }
    SpgWgradParams w; memset(&w, 0, sizeof(w));
    w.a = cur; w.b = input_operand(pl, sg, false, k, clouds, stnT); w.M = (int)pl.M; w.N = l.cout; w.K = l.cin;
    w.fold = pending; memset(&pending, 0, sizeof(pending));
    w.allow_lowp = 1;      // the opt-in precision modes act on the PointNet convolutions only
    // This weight gradient is a LEAF (nobody reads dW before the optimiser) that used to stand in front of the data gradient the
    // rest of the backward waits for.  As leaves (spg_gemm.h) slices of it leave later, next to the STN head's few-row launches.
    // It then applies its layer's BatchNorm-backward constants from a PRIVATE copy (finished in its own prologue from the same
    // fixed-point sums: same bits), because `consts` is rewritten by the layers in between; the data gradient below finishes the
    // shared copy itself, as it does inside a grouped launch.
    bool leaf = false;
    if (leaves == 1 && pl.fold && w.fold.slots != nullptr && (k == 0 || k + 1 == (int)sg.convs.size())) {
      float* cl = s.consts_leaf[k == 0 ? 0 : 1];
      SpgWgradParams wl = w;
      const int C = w.fold.C;
      wl.fold.consts = cl;
      wl.a.c0 = cl; wl.a.c1 = cl + C; wl.a.c2 = cl + 2 * C; wl.a.c3 = cl + 3 * C;
      // pooled layer: 8.4 GFLOP -> 6 slices of ~15 us; first layer (K = 14: bandwidth, ~25 us) -> 2
      leaf = cur.c0 == s.consts && spg_queue_wgrad_leaf(rq, wl, l.dW, k == 0 ? 2 : 6, st);
    }
    const SpgBnFoldBwd fold_for_dgrad = leaf ? w.fold : SpgBnFoldBwd{};
    if (!leaf) SPG_TRY(spg_queue_wgrad(rq, w, l.dW, st));
    if (l.db) SPG_TRY(zero_async(l.db, l.cout, st));
    if (k > 0) {
      Layer& prod = pl.L[sg.convs[k - 1]];
      float* out = dz[flip]; flip ^= 1;
      SpgGemmParams g; memset(&g, 0, sizeof(g));
      g.a = cur; g.W = l.W; g.ldw = l.cin; g.w_red = 1;
      g.M = (int)pl.M; g.N = l.cin; g.K = l.cout; g.rows_per_tile = pl.P;
      if (l.Wb_t != nullptr) { g.Wb = l.Wb_t; g.ldwb = spg_split_ld(l.cout); g.wb_part_bytes = (long)l.cin * g.ldwb * 2; }
      g.epi = SPG_EPI_BWD; g.Y = out; g.ldy = l.cin; g.Yp = prod.y; g.ldyp = prod.ldy;
      g.ms = prod.s; g.mt = prod.t; g.mask_relu = 1; g.n_mask = prod.cout;
      g.mmean = prod.mean; g.mrstd = prod.rstd; g.stat = s.stat;
      const bool foldk = pl.fold && !spg_gemm_bwd_stats_want_partials(g);
      if (foldk) { g.stat = nullptr; g.stat_slots = prod.slots_bwd; }
      g.fold_bwd = fold_for_dgrad;      // (the layer's weight gradient, which otherwise finishes the constants first, leaves later)
      int nparts = 0;
      SPG_TRY(spg_launch_gemm(g, st, &nparts));
      if (foldk) pending = fold_bwd_of(pl, prod, pl.M, s.consts);
      else
      SPG_TRY(spg_launch_bn_bwd_finalize(s.stat, nparts, l.cin, pl.M, prod.cout, prod.s, prod.mean, prod.rstd, s.consts,
                                         prod.dgamma, prod.dbeta, s.fin, st));
      cur = op_bnbwd(out, prod.y, prod.ldy, s.consts, prod.cout);
    } else if (want_dxy) {
      // gradient wrt the transformed xy only (learning/pointnet.py:123-124): 2 output columns
      SpgGemmParams g; memset(&g, 0, sizeof(g));
      g.a = cur; g.W = l.Wpad ? l.Wpad : l.W; g.ldw = l.Wpad ? l.ldw : l.cin; g.w_red = 1;
      g.M = (int)pl.M; g.N = 2; g.K = l.cout; g.rows_per_tile = pl.P;
      g.epi = SPG_EPI_BWD; g.Y = s.dxy; g.ldy = 2;
      g.fold_bwd = fold_for_dgrad;
      SPG_TRY(spg_launch_gemm(g, st));
    }
  }
  return 0;
}

void bind_grads(Plan& pl, void* const* grads) {
  for (size_t i = 0; i < pl.L.size(); ++i) {
    void* const* g = grads + 6 * i;
    pl.L[i].dW = (float*)g[0]; pl.L[i].db = (float*)g[1]; pl.L[i].dgamma = (float*)g[2]; pl.L[i].dbeta = (float*)g[3];
  }
}

}  // namespace

extern "C" int spg_pointnet_num_layers(const spg_pointnet_cfg* cfg) { return cfg ? num_layers(*cfg) : -1; }

extern "C" size_t spg_pointnet_workspace_bytes(const spg_pointnet_cfg* cfg, int B, int training) {
  Plan pl;
  if (make_plan(cfg, B, training, nullptr, nullptr, nullptr, pl) != 0) return 0;
  return pl.bytes;
}

extern "C" int spg_pointnet_forward(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                                    const void* const* params, float* emb, void* workspace, int training,
                                    int bn_update_times, void* stream) {
  return spg_pointnet_forward_ext(cfg, B, clouds, clouds_global, nullptr, params, emb, workspace, training, bn_update_times, stream);
}

// spg_train_step (spg_step.hip), per thread: `clean` -- a train-mode forward does NOT clear the statistics slots (the caller
// vouches that the previous step on the same workspace left them zero); `clear_at_end` -- the backward clears them as a job of
// its final batched reduction (a reduction over ZERO partials writes zeros): the last launch of the step, behind every producer
// and consumer of the slots
static thread_local bool g_slots_clean = false, g_slots_clear_at_end = false;
void spg_pointnet_set_step_flags(bool clean, bool clear_at_end) { g_slots_clean = clean; g_slots_clear_at_end = clear_at_end; }

extern "C" int spg_pointnet_forward_ext(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                                        const float* ext_transform, const void* const* params, float* emb, void* workspace,
                                        int training, int bn_update_times, void* stream) {
  SPG_CHECK_ARG(clouds && params && emb && workspace, "null pointer");
  SPG_CHECK_ARG(ext_transform == nullptr || cfg->nfeat_stn == 0, "an external transform replaces the inner STN (nfeat_stn must be 0)");
  hipStream_t st = (hipStream_t)stream;
  Plan pl;
  SPG_TRY(make_plan(cfg, B, training, workspace, params, emb, pl));
  SPG_CHECK_ARG(pl.cfg.nfeat_global == 0 || clouds_global != nullptr, "clouds_global is required");
  pl.main.extra = clouds_global;
  if (!pl.training) {        // eval mode: the BatchNorm constants of every layer in one launch
    SpgBnEvalBatch eb;
    for (Layer& l : pl.L)
      if (l.bn) {
        SPG_CHECK_ARG(eb.njobs < SPG_BN_EVAL_MAX_JOBS, "too many BatchNorm layers");
        eb.jobs[eb.njobs++] = SpgBnEvalJob{l.cout, l.gamma, l.beta, l.rm, l.rv, l.s, l.t};
      }
    SPG_TRY(spg_launch_bn_eval_batch(eb, pl.cfg.bn_eps, st));
  }
  if (pl.training && spg_gemm_precision() != 0) {      // opt-in bf16 / split-bf16 MFMA: bf16 copies of the conv weights, one launch
    SpgSplitBatch sb;
    for (Layer& l : pl.L)
      if (l.Wb_f != nullptr) {
        SPG_CHECK_ARG(sb.njobs < SPG_SPLIT_MAX_JOBS, "too many layers");
        sb.jobs[sb.njobs++] = SpgSplitJob{l.W, l.cin, l.cout, l.cin, l.Wb_f, l.Wb_t};
      }
    SPG_TRY(spg_launch_split_weights(sb, st));
  }
  // train mode: BatchNorm statistics travel as fixed-point slots from each producer GEMM to its consumer (spg_gemm.h) -- no
  // finalize launches; not with synchronised BatchNorm (the ranks' all-reduce sits between producer and consumer)
  // (every tile contributes at most 4 wave partials per channel: far below the slots' capacity up to ~500 k superpoints)
  pl.fold = pl.training && !spg_sync_bn_active() && !spg_tune_get(SPG_TUNE_NO_BN_FOLD) && 4L * B * spg_slot_sync_world() <= SPG_FOLD_MAX_CONTRIBUTIONS;
  SPG_CHECK_ARG(!(pl.training && spg_slot_sync_active()) || pl.fold, "slot-synchronised BatchNorm needs the statistics slots (spg_tune key 10 off, batch within the slots' capacity)");
  if (pl.training && !g_slots_clean) {      // always in train mode: the backward decides about its own slots independently (they are cleared here too)
    hipError_t me = hipMemsetAsync(pl.slots_all, 0, pl.slots_words * sizeof(unsigned long long), st);
    if (me != hipSuccess) { spg_set_error("hipMemsetAsync: %s", hipGetErrorString(me)); return (int)me; }
  }
  const float* stnT = ext_transform;      // [B, 4] = T - I of an externally evaluated STN (LocalCloudEmbedder), or null
  if (pl.has_stn) {
    SPG_TRY(forward_segment(pl, pl.stn, clouds, nullptr, bn_update_times, st));
    stnT = pl.L[pl.stn.fcs.back()].y;
  }
  SPG_TRY(forward_segment(pl, pl.main, clouds, stnT, bn_update_times, st));
  return 0;
}

// debug / test helper: byte offset inside the forward workspace of a layer's buffers
//   what: 0 raw output y, 1 BN scale s, 2 BN shift t, 3 batch mean, 4 batch rstd;
//   layer == -1 / -2: STN / main segment; what 0: pooled [B, ldpool] (selected raw max/min + globals), what 1: aidx [B, ldpool]
//   (int32: the point that holds the pooled value), what 2: ldpool itself (not an offset).
extern "C" long spg_pointnet_debug_offset(const spg_pointnet_cfg* cfg, int B, int training, int layer, int what) {
  Plan pl;
  char* fake = (char*)(uintptr_t)4096;
  if (make_plan(cfg, B, training, fake, nullptr, (float*)(uintptr_t)8, pl) != 0) return -1;
  const void* p = nullptr;
  if (layer == -1 || layer == -2) {
    const Segment& sg = layer == -1 ? pl.stn : pl.main;
    if (layer == -1 && !pl.has_stn) return -1;
    if (what == 2) return sg.ldpool;
    p = what == 1 ? (const void*)sg.aidx : (const void*)sg.pooled;
  }
  else if (layer >= 0 && layer < (int)pl.L.size()) {
    const Layer& l = pl.L[layer];
    p = what == 0 ? (const void*)l.y : what == 1 ? l.s : what == 2 ? l.t : what == 3 ? l.mean : l.rstd;
    if (what == 0 && l.y == (float*)(uintptr_t)8) return -1;
  }
  if (p == nullptr) return -1;
  return (long)((const char*)p - fake);
}

extern "C" size_t spg_pointnet_bwd_workspace_bytes(const spg_pointnet_cfg* cfg, int B) {
  Plan pl;
  if (make_plan(cfg, B, 1, nullptr, nullptr, nullptr, pl) != 0) return 0;
  BwdScratch s;
  carve_bwd(pl, nullptr, s);
  return s.bytes;
}

extern "C" int spg_pointnet_backward(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                                     const void* const* params, const float* grad_emb, void* const* grads,
                                     void* workspace, void* bwd_workspace, void* stream) {
  return spg_pointnet_backward_ext(cfg, B, clouds, clouds_global, nullptr, params, grad_emb, grads, nullptr, nullptr, workspace,
                                   bwd_workspace, stream);
}

extern "C" int spg_pointnet_backward_ext(const spg_pointnet_cfg* cfg, int B, const float* clouds, const float* clouds_global,
                                         const float* ext_transform, const void* const* params, const float* grad_emb,
                                         void* const* grads, float* grad_transform, float* grad_global, void* workspace,
                                         void* bwd_workspace, void* stream) {
  SPG_CHECK_ARG(clouds && params && grad_emb && grads && workspace && bwd_workspace, "null pointer");
  SPG_CHECK_ARG((ext_transform == nullptr && grad_transform == nullptr) || cfg->nfeat_stn == 0, "external transform with an inner STN");
  SPG_CHECK_ARG(grad_transform == nullptr || ext_transform != nullptr, "grad_transform needs the external transform");
  hipStream_t st = (hipStream_t)stream;
  Plan pl;
  // the forward wrote the last fc output to `emb`; it is not needed by the backward, so pass a dummy
  SPG_TRY(make_plan(cfg, B, 1, workspace, params, (float*)grad_emb /*unused as y*/, pl));
  pl.main.extra = clouds_global;
  pl.fold = !spg_sync_bn_active() && !spg_tune_get(SPG_TUNE_NO_BN_FOLD) && 4L * B * spg_slot_sync_world() <= SPG_FOLD_MAX_CONTRIBUTIONS;
  bind_grads(pl, grads);
  BwdScratch s;
  carve_bwd(pl, bwd_workspace, s);
  const float* stnT = pl.has_stn ? pl.L[pl.stn.fcs.back()].y : ext_transform;
  const int cout = pl.L[pl.main.fcs.back()].cout;
  SpgReduceQueue rq;
  rq.arena = s.work; rq.arena_floats = s.work_floats;
  s.grad_global = grad_global;
  struct LeafGuard { ~LeafGuard() { spg_leaf_clear(); } } leaf_guard;      // no leaf survives this call (its buffers are the caller's)
  SPG_CHECK_ARG(spg_leaf_pending() == 0, "leaves of another call are pending on this thread");
  bool dT_done = false;
  float* dT_target = grad_transform != nullptr ? grad_transform : (pl.has_stn ? s.dT : nullptr);
  SPG_TRY(backward_segment(pl, pl.main, s, rq, op_ident(grad_emb, cout), clouds, stnT, pl.has_stn || grad_transform != nullptr, st, false,
                           pl.has_stn ? 1 : 0, dT_target, &dT_done));
  if (grad_transform != nullptr && !dT_done)      // gradient wrt the external 2x2 transforms (learning/pointnet.py:196-198)
    SPG_TRY(spg_launch_stn_dT(clouds, pl.cfg.nfeat, pl.P, B, s.dxy, 2, grad_transform, st));
  if (pl.has_stn) {
    if (!dT_done) SPG_TRY(spg_launch_stn_dT(clouds, pl.cfg.nfeat, pl.P, B, s.dxy, 2, s.dT, st));
    SPG_TRY(backward_segment(pl, pl.stn, s, rq, op_ident(s.dT, 4), clouds, nullptr, false, st, true, 2));
  }
  SPG_TRY(spg_leaf_drain(st));      // (leaves no head took along: a launch of their own)
  if (g_slots_clear_at_end && pl.slots_all != nullptr && pl.slots_words > 0) {
    SPG_CHECK_ARG(2 * pl.slots_words < (size_t)INT_MAX, "statistics slots too large for one reduction job");
    SpgReduceJob j;
    j.partial = reinterpret_cast<const float*>(pl.slots_all); j.out = reinterpret_cast<float*>(pl.slots_all);
    j.nsplit = 0; j.n = (int)(2 * pl.slots_words);
    spg_reduce_defer(j);
  }
  return spg_flush_reduce(rq, st);      // ONE launch sums the split partials of all weight / bias gradients
}
