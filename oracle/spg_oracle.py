"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain torch-CPU / numpy arithmetic, no custom kernels) of the
superpoint-graph learning hot path of loicland/superpoint_graph:

    PointNet superpoint embedding  ->  edge-conditioned graph convolution (ECC)
    with GRU update  ->  linear classifier  ->  weighted cross entropy,
    forward and backward.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / CPU baseline.  The product
(`superpoint_graph_amd/`) never imports it and fails loudly when the HIP
extension is missing.

Every function cites the reference file:line it restates (paths relative to the
upstream repository root).  Parity pinning: `oracle/validate_against_reference.py`
imports the upstream modules (in the build container, where /root/reference
exists) and checks every function below against them on seeded inputs; the same
script writes the golden vectors in `tests/golden/` that travel to the GPU box.
The upstream repository holds no golden vectors of its own (only the gradcheck /
shard-invariance property tests in learning/ecc/test_GraphConvModule.py, which
are restated in tests/test_oracle.py), so the pin is "outputs of the reference
itself, run here".

All parameters are addressed by the reference's `state_dict` key names
(e.g. 'ptn.stn.convs.0.weight', 'ecc.0._cell.weight_ih'), so one dict feeds the
reference modules, this oracle and the HIP modules alike.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

BN_EPS = 1e-5        # nn.BatchNorm1d default, learning/pointnet.py:31 / graphnet.py:29
BN_MOMENTUM = 0.1    # nn.BatchNorm1d default
IN_EPS = 1e-5        # nn.InstanceNorm1d(1, eps=1e-5), learning/modules.py:213-214


# --------------------------------------------------------------------------------------
# model description (mirrors the CLI arguments of learning/main.py:43-113)
# --------------------------------------------------------------------------------------
@dataclass
class ModelSpec:
    """The subset of learning/main.py's arguments that defines the network
    (create_model, learning/main.py:414-431)."""
    model_config: str = 'gru_10_0,f_13'
    node_feats: int = 14                 # dbinfo['node_feats'] = len(pc_attribs)
    edge_feats: int = 13                 # dbinfo['edge_feats']
    ptn_widths: Sequence[Sequence[int]] = ((64, 64, 128, 128, 256), (256, 64, 32))
    ptn_widths_stn: Sequence[Sequence[int]] = ((64, 64, 128), (128, 64))
    ptn_nfeat_stn: int = 14
    ptn_prelast_do: float = 0.0
    fnet_widths: Sequence[int] = (32, 128, 64)
    fnet_llbias: int = 0
    fnet_orthoinit: int = 1
    fnet_bnidx: int = 2
    ptn_npts: int = 128


@dataclass
class RnnEccSpec:
    """One 'gru_R[_vv][_layernorm][_ingate][_catall]' token (learning/graphnet.py:66-82)."""
    kind: str
    nrepeats: int
    vv: bool
    layernorm: bool
    ingate: bool
    cat_all: bool


def parse_model_config(config: str, nfeat: int):
    """learning/graphnet.py:44-84.  Returns a list of (module_index, kind, payload)."""
    out = []
    for d, conf in enumerate(config.split(',')):
        conf = conf.strip().split('_')
        if conf[0] == 'f':
            out.append((d, 'f', (nfeat, int(conf[1]))))
            nfeat = int(conf[1])
        elif conf[0] in ('gru', 'lstm'):
            nrepeats = int(conf[1])
            vv = bool(int(conf[2])) if len(conf) > 2 else True
            layernorm = bool(int(conf[3])) if len(conf) > 3 else True
            ingate = bool(int(conf[4])) if len(conf) > 4 else True
            cat_all = bool(int(conf[5])) if len(conf) > 5 else True
            out.append((d, conf[0], (nfeat, RnnEccSpec(conf[0], nrepeats, vv, layernorm, ingate, cat_all))))
            if cat_all:
                nfeat *= nrepeats + 1
        elif conf[0] == 'r':
            out.append((d, 'r', None))
        elif conf[0] == 'b':
            out.append((d, 'b', (nfeat, len(conf) == 1)))
        elif conf[0] == 'd':
            out.append((d, 'd', float(conf[1])))
        elif len(conf[0]) > 0:
            raise NotImplementedError('Unknown module: ' + conf[0])
    return out


# --------------------------------------------------------------------------------------
# integer / index work (bit-exact contract)
# --------------------------------------------------------------------------------------
def set_batch(edge_lists: Sequence[np.ndarray], vcounts: Sequence[int],
              edge_feats: Sequence[np.ndarray]):
    """GraphConvInfo.set_batch, learning/ecc/GraphConvInfo.py:33-69, without igraph.

    edge_lists[g]: int array [E_g, 2] of (source, target) pairs in igraph edge order,
    vcounts[g]: number of vertices, edge_feats[g]: float32 [E_g, Fe] (edge attribute 'f').
    Returns idxn i64[E], degs i64[N], edgefeats f32[E,Fe], edge_indexes i64[2,E].
    """
    p = 0
    idxn, degrees, edge_indexes, feats = [], [], [], []
    for E, n, f in zip(edge_lists, vcounts, edge_feats):
        E = np.asarray(E).reshape(-1, 2)
        idx = E[:, 1].argsort()                       # :50 sort by target (numpy default kind)
        idxn.append(p + E[idx, 0])                    # :52
        feats.append(np.asarray(f)[idx])              # :53-55 edge attrs in that order
        degrees.append(np.bincount(E[:, 1], minlength=n).astype(np.int64))  # :56 indegree
        edge_indexes.append(np.asarray(p + E[idx]))   # :57
        p += n                                        # :58
    idxn = np.concatenate(idxn).astype(np.int64)
    degs = np.concatenate(degrees).astype(np.int64)
    edgefeats = np.concatenate(feats).astype(np.float32)
    edge_indexes = np.concatenate(edge_indexes).T.astype(np.int64)
    return idxn, degs, edgefeats, edge_indexes


def get_edge_shards(degs, edge_mem_limit):
    """learning/ecc/utils.py:56-69."""
    d = np.asarray(degs)
    cs = np.cumsum(d)
    cse = cs // edge_mem_limit
    _, cse_i, cse_c = np.unique(cse, return_index=True, return_counts=True)
    shards = []
    for b in range(len(cse_i)):
        numd = cse_c[b]
        nume = (cs[-1] if b == len(cse_i) - 1 else cs[cse_i[b + 1] - 1]) - cs[cse_i[b]] + d[cse_i[b]]
        shards.append((int(numd), int(nume)))
    return shards


def csr_by_target(degs: np.ndarray) -> np.ndarray:
    """rowptr i64[N+1]: start of each destination node's edge segment
    (cuda_kernels.py:63 `cslengths[i] - lengths[i]`)."""
    rp = np.zeros(len(degs) + 1, dtype=np.int64)
    np.cumsum(degs, out=rp[1:])
    return rp


def csr_by_source(idxn: np.ndarray, n_nodes: int):
    """Reverse CSR used by the atomic-free grad_input scatter (restates the index_add_
    of learning/ecc/GraphConvModule.py:146 as a gather): for each source node j the
    list of edge ids e with idxn[e]==j, in increasing e.  Stable counting sort."""
    order = np.argsort(idxn, kind='stable').astype(np.int64)
    counts = np.bincount(idxn, minlength=n_nodes).astype(np.int64)
    rp = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rp[1:])
    return rp, order


def edge_targets(degs: np.ndarray) -> np.ndarray:
    """Destination node of every edge (edges are sorted by target)."""
    return np.repeat(np.arange(len(degs), dtype=np.int64), degs)


# --------------------------------------------------------------------------------------
# ECC aggregate (GraphConvFunction)
# --------------------------------------------------------------------------------------
def ecc_forward(x: torch.Tensor, w: torch.Tensor, idxn: torch.Tensor, degs: torch.Tensor,
                idxe: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GraphConvFunction.forward, learning/ecc/GraphConvModule.py:44-94 (+ the segment mean of
    cuda_kernels.py:55-86).  out[i] = mean_{e in in(i)} x[idxn[e]] @ W_e   (matrix, w: [E,in,out])
    or x[idxn[e]] * w_e (vector, w: [E,nc]); rows with deg 0 are exactly 0.
    The result is invariant to edge_mem_limit sharding, so no shards here."""
    n = degs.numel()
    dst = torch.repeat_interleave(torch.arange(n), degs)
    sel = x.index_select(0, idxn)                                  # :66
    ww = w if idxe is None else w.index_select(0, idxe)            # :68-71
    if w.dim() == 3:
        prod = torch.bmm(sel.unsqueeze(1), ww).squeeze(1)          # :38
    else:
        prod = sel * ww                                            # :41
    out = torch.zeros(n, prod.shape[1], dtype=x.dtype)
    out.index_add_(0, dst, prod)
    return out / degs.clamp(min=1).to(x.dtype).unsqueeze(1)        # :81-88 mean / zero


def ecc_backward(x, w, grad_out, idxn, degs, idxe=None):
    """GraphConvFunction.backward, learning/ecc/GraphConvModule.py:97-152.
    (The reference's own matrix-mode backward raises on torch>=1.5 at :146 -- shape-strict
    index_add_ -- so this restatement IS the matrix-mode checker; it is validated by fp64
    gradcheck on the reference test's fixture and against the reference vector-mode backward.)"""
    n = degs.numel()
    dst = torch.repeat_interleave(torch.arange(n), degs)
    g = grad_out.index_select(0, dst) / degs.index_select(0, dst).to(x.dtype).unsqueeze(1)  # :108-121
    sel = x.index_select(0, idxn)
    ww = w if idxe is None else w.index_select(0, idxe)
    if w.dim() == 3:
        gw_e = sel.unsqueeze(2) * g.unsqueeze(1)                   # :126-133  x^T (x) g
        gsel = torch.bmm(g.unsqueeze(1), ww.transpose(1, 2)).squeeze(1)   # :135-144
    else:
        gw_e = sel * g
        gsel = g * ww
    if idxe is None:
        gw = gw_e
    else:
        gw = torch.zeros_like(w)
        gw.index_add_(0, idxe, gw_e)                               # :130
    gx = torch.zeros_like(x)
    gx.index_add_(0, idxn, gsel)                                   # :146
    return gx, gw


class EccFunction(torch.autograd.Function):
    """autograd wrapper of ecc_forward/ecc_backward (same signature as the reference
    GraphConvFunction.apply, learning/ecc/GraphConvModule.py:44)."""

    @staticmethod
    def forward(ctx, input, weights, in_channels, out_channels, idxn, idxe, degs, degs_gpu=None, edge_mem_limit=1e20):
        ctx.save_for_backward(input, weights)
        ctx.meta = (idxn, idxe, degs)
        return ecc_forward(input, weights, idxn, degs, idxe)

    @staticmethod
    def backward(ctx, grad_output):
        input, weights = ctx.saved_tensors
        idxn, idxe, degs = ctx.meta
        gx, gw = ecc_backward(input, weights, grad_output, idxn, degs, idxe)
        return gx, gw, None, None, None, None, None, None, None


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def batch_norm(x: torch.Tensor, prefix: str, P: Dict[str, torch.Tensor], training: bool,
               stats_out: Optional[dict] = None) -> torch.Tensor:
    """nn.BatchNorm1d on [M, C] or [B, C, L] (channel = dim 1).  Train: biased batch variance for the
    normalisation; `stats_out[prefix] = (mean, unbiased_var, count)` lets the caller apply the
    running-stat update (momentum 0.1, unbiased variance)."""
    red = (0,) if x.dim() == 2 else (0, 2)
    if training:
        if stats_out is not None:
            with torch.no_grad():
                m = x.numel() // x.shape[1]
                mean = x.mean(red)
                var = x.var(red, unbiased=False)
                stats_out[prefix] = (mean, var * (m / max(m - 1, 1)), m)
        # the op nn.BatchNorm1d.forward itself calls (learning/pointnet.py:31, graphnet.py:29): batch statistics, and torch's
        # native backward.  (Until round 3 this was the written-out formula differentiated by autograd: same forward, but
        # at 128 000 rows its fp32 gradients were 6e-3..3e-2 off the fp64 truth where the reference's are 1e-5..7e-4 --
        # measured in oracle/validate_against_reference.py::check_baseline_size -- and twice as slow.)
        return torch.nn.functional.batch_norm(x, None, None, P[prefix + '.weight'].to(x.dtype), P[prefix + '.bias'].to(x.dtype),
                                              True, 0.0, BN_EPS)
    shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
    mean = P[prefix + '.running_mean'].to(x.dtype).view(shape)
    var = P[prefix + '.running_var'].to(x.dtype).view(shape)
    return (x - mean) / torch.sqrt(var + BN_EPS) * P[prefix + '.weight'].to(x.dtype).view(shape) + P[prefix + '.bias'].to(x.dtype).view(shape)


def apply_running_stats(P: Dict[str, torch.Tensor], stats: dict, times: int = 1):
    """BatchNorm running-stat update; `times=2` reproduces run_full_monger's double forward
    (learning/pointnet.py:167,173: num_batches_tracked += 2 per training step)."""
    for prefix, (mean, uvar, _m) in stats.items():
        for _ in range(times):
            P[prefix + '.running_mean'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.to(P[prefix + '.running_mean'].dtype))
            P[prefix + '.running_var'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * uvar.to(P[prefix + '.running_var'].dtype))
            P[prefix + '.num_batches_tracked'] += 1


def _lin(x, P, prefix, bias=True):
    w = P[prefix + '.weight'].to(x.dtype)
    if w.dim() == 3:               # Conv1d(k=1) weight [Cout, Cin, 1]
        w = w.squeeze(2)
    y = x @ w.t()
    if bias and (prefix + '.bias') in P:
        y = y + P[prefix + '.bias'].to(x.dtype)
    return y


# Decision hooks (tests only): the two non-smooth operations of PointNet -- ReLU and the max-pool over the points -- are
# DECISIONS (which side of zero, which point wins).  Two fp32 implementations that agree to round-off can still take a
# different decision on a near-tie, and then their gradients differ by far more than round-off.  `rec` (a dict) receives,
# per BatchNorm prefix, the value the ReLU saw -- point layers as [B, Pn, C] (row b*Pn + p of the [M, C] matrices the HIP
# kernels use), FC layers as [B, C] -- and, per '<segment>.pool', the [B, Pn, C] tensor the max-pool saw; `dec` forces
# decisions taken elsewhere (ReLU: bool mask [B*Pn, C] / [B, C]; pool: int64 [B, C] point indices), so that a backward
# pass can be compared with the decisions held equal.
def _relu_dec(v, key, dec, rec):
    pts = v.dim() == 3                                   # [B, C, Pn]
    if rec is not None:
        rec[key] = v.detach().permute(0, 2, 1) if pts else v.detach()
    if dec is not None and key in dec:
        m = dec[key]
        if pts:
            m = m.view(v.shape[0], v.shape[2], v.shape[1]).permute(0, 2, 1)
        return v * m.to(v.dtype)
    return torch.relu(v)


def _pool_dec(h, key, dec, rec):
    """h [B, C, Pn] -> [B, C]: max over the points (nnf.max_pool1d(input, input.size(2)).squeeze(2))."""
    if rec is not None:
        rec[key] = h.detach().permute(0, 2, 1)
    if dec is not None and key in dec:
        return h.gather(2, dec[key].unsqueeze(2)).squeeze(2)
    return torch.nn.functional.max_pool1d(h, h.size(2)).squeeze(2)


def _conv1(x, P, prefix):
    """nn.Conv1d(cin, cout, 1) on [B, C, Pn] (learning/pointnet.py:30,86)."""
    w = P[prefix + '.weight'].to(x.dtype)
    return torch.nn.functional.conv1d(x, w if w.dim() == 3 else w.unsqueeze(2), P[prefix + '.bias'].to(x.dtype))


def stn_forward(clouds_stn: torch.Tensor, spec: ModelSpec, P, training: bool, stats=None, pfx='ptn.stn', dec=None, rec=None):
    """STNkD.forward, learning/pointnet.py:55-61.  clouds_stn: [B, nfeat_stn, Pn] -> T [B,2,2].  Same op sequence and
    tensor layout as the reference ([B, C, Pn] through Conv1d / BatchNorm1d): at 10^5..10^6 points the fp32 statistics of
    a [M, C] re-layout differ measurably from the reference's (oracle/validate_against_reference.py::check_local_embedder)."""
    h = clouds_stn
    for i, _w in enumerate(spec.ptn_widths_stn[0]):               # :27-37 (conv, bn, relu) triples
        h = _conv1(h, P, f'{pfx}.convs.{3 * i}')
        h = _relu_dec(batch_norm(h, f'{pfx}.convs.{3 * i + 1}', P, training, stats), f'{pfx}.convs.{3 * i + 1}', dec, rec)
    h = _pool_dec(h, f'{pfx}.pool', dec, rec)                     # :58 max_pool1d over points
    for i, _w in enumerate(spec.ptn_widths_stn[1]):               # :39-49
        h = _lin(h, P, f'{pfx}.fcs.{3 * i}')
        h = _relu_dec(batch_norm(h, f'{pfx}.fcs.{3 * i + 1}', P, training, stats), f'{pfx}.fcs.{3 * i + 1}', dec, rec)
    h = _lin(h, P, f'{pfx}.proj')                                 # :60
    return h.view(-1, 2, 2) + torch.eye(2, dtype=h.dtype).unsqueeze(0)   # :61


def pointnet_forward(clouds: torch.Tensor, clouds_global: torch.Tensor, spec: ModelSpec, P,
                     training: bool, stats=None, pfx='ptn', dec=None, rec=None):
    """PointNet.forward, learning/pointnet.py:120-133.  clouds [B,F,Pn], clouds_global [B] or [B,G] -> [B,D]."""
    B, F, Pn = clouds.shape
    if spec.ptn_nfeat_stn > 0:
        T = stn_forward(clouds[:, :spec.ptn_nfeat_stn, :], spec, P, training, stats, pfx + '.stn', dec, rec)   # :122
        xy = torch.bmm(clouds[:, :2, :].transpose(1, 2), T).transpose(1, 2)                         # :123
        clouds = torch.cat([xy, clouds[:, 2:, :]], 1)                                                # :124
    h = clouds
    for i, _w in enumerate(spec.ptn_widths[0]):                   # :83-96
        h = _conv1(h, P, f'{pfx}.convs.{3 * i}')
        h = _relu_dec(batch_norm(h, f'{pfx}.convs.{3 * i + 1}', P, training, stats), f'{pfx}.convs.{3 * i + 1}', dec, rec)
    h = _pool_dec(h, f'{pfx}.pool', dec, rec)                     # :127
    if clouds_global is not None:
        h = torch.cat([h, clouds_global.view(B, -1).to(h.dtype)], 1)     # :128-132
    nfc = len(spec.ptn_widths[1])
    idx = 0
    for i in range(nfc):                                          # :98-118
        h = _lin(h, P, f'{pfx}.fcs.{idx}')
        idx += 1
        if i < nfc - 1:                                           # last_ac=False
            h = _relu_dec(batch_norm(h, f'{pfx}.fcs.{idx}', P, training, stats), f'{pfx}.fcs.{idx}', dec, rec)
            idx += 2
        if i == nfc - 2 and spec.ptn_prelast_do > 0:
            idx += 1                                              # nn.Dropout slot (identity here: p must be 0 for parity runs)
    return h


def cloud_embed(clouds_flag: torch.Tensor, clouds: torch.Tensor, clouds_global: torch.Tensor,
                spec: ModelSpec, P, training: bool, stats=None, dec=None, rec=None):
    """CloudEmbedder.run_full / run_full_monger, learning/pointnet.py:147-180: PointNet over the
    valid superpoints, scattered into a zero [N_total, D] matrix (invalid rows exactly 0)."""
    idx_valid = torch.nonzero(clouds_flag.eq(0)).squeeze(1)       # :149 (0-d squeeze gotcha avoided)
    out = pointnet_forward(clouds, clouds_global, spec, P, training, stats, dec=dec, rec=rec)
    desc = torch.zeros(clouds_flag.shape[0], out.shape[1], dtype=out.dtype)
    return desc.index_copy(0, idx_valid, out), idx_valid          # :178-179


LOCAL_CHUNK = 2 ** 16 - 1      # learning/pointnet.py:193 (a cuDNN workaround in the reference; it also fixes the BatchNorm batches)


def local_cloud_embed(clouds: torch.Tensor, clouds_global: torch.Tensor, spec: ModelSpec, P, training: bool,
                      nfeat_stn: int = 2, stn_as_global: bool = True, update_running: bool = False, dec=None, rec=None):
    """LocalCloudEmbedder.run_batch, learning/pointnet.py:189-205 (the supervised partition's embedder,
    supervized_partition.py:411-421): a stand-alone STN (state_dict prefix 'stn') on the first `nfeat_stn` channels, the 2x2
    transform applied to xy, the transform optionally appended to the global features, a PointNet WITHOUT inner STN
    (prefix 'ptn'; `spec.ptn_nfeat_stn` must be 0), L2 normalisation.  clouds [n, F, k], clouds_global [n, G] -> [n, D].
    The reference evaluates chunks of 2^16 - 1 clouds (:193-198, :204-206): in training mode every chunk is its own
    BatchNorm batch (own statistics, own running-stat update) -- restated here because it is observable.
    update_running: apply the running-stat updates to P chunk by chunk, in the reference's order (all STN chunks, then all
    PointNet chunks)."""
    assert spec.ptn_nfeat_stn == 0, 'the local embedder uses a PointNet without inner STN'
    n = clouds.shape[0]
    bounds = [(a, min(n, a + LOCAL_CHUNK)) for a in range(0, n, LOCAL_CHUNK)]      # :193-194 (n_batches = int((n-1)/batch_size))

    def chunked(fn):
        outs = []
        for a, b in bounds:
            stats = {} if (training and update_running) else None
            outs.append(fn(a, b, stats))
            if stats:
                apply_running_stats(P, stats, 1)
        return torch.cat(outs)

    if nfeat_stn > 0:
        T = chunked(lambda a, b, st: stn_forward(clouds[a:b, :nfeat_stn, :], spec, P, training, st, 'stn', dec, rec))   # :196-198
        xy = torch.bmm(clouds[:, :2, :].transpose(1, 2), T).transpose(1, 2)        # :199
        clouds = torch.cat([xy, clouds[:, 2:, :]], 1)                              # :200
        if stn_as_global:
            clouds_global = torch.cat([clouds_global, T.reshape(-1, 4)], 1)        # :201-202
    out = chunked(lambda a, b, st: pointnet_forward(clouds[a:b], clouds_global[a:b], spec, P, training, st, 'ptn', dec, rec))  # :204-206
    return torch.nn.functional.normalize(out)                                      # :207


def fnet_forward(edgefeats: torch.Tensor, spec: ModelSpec, nout: int, P, training: bool, stats=None, pfx='ecc.0._fnet', dec=None, rec=None):
    """create_fnet product, learning/graphnet.py:17-34: Linear/ReLU stack with one BatchNorm at
    `bnidx`, last Linear without ReLU (bias iff llbias)."""
    widths = [spec.edge_feats] + list(spec.fnet_widths) + [nout]
    h = edgefeats
    idx = 0
    for k in range(len(widths) - 2):
        h = _lin(h, P, f'{pfx}.{idx}')
        idx += 1
        if spec.fnet_bnidx == k:
            h = batch_norm(h, f'{pfx}.{idx}', P, training, stats)
            idx += 1
        h = _relu_dec(h, f'{pfx}.relu{k}', dec, rec)             # decision hook key: the k-th ReLU of the filter network
        idx += 1
    h = _lin(h, P, f'{pfx}.{idx}', bias=bool(spec.fnet_llbias))
    idx += 1
    if spec.fnet_bnidx == len(widths) - 1:
        h = batch_norm(h, f'{pfx}.{idx}', P, training, stats)
    return h


def _row_norm(g: torch.Tensor) -> torch.Tensor:
    """InstanceNorm1d(1) on g.unsqueeze(1): per-row (x-mean)/sqrt(var_biased+eps), no affine
    (learning/modules.py:218-222)."""
    mu = g.mean(1, keepdim=True)
    var = g.var(1, unbiased=False, keepdim=True)
    return (g - mu) / torch.sqrt(var + IN_EPS)


def gru_cell_ex(inp, hidden, P, pfx='ecc.0._cell', layernorm=True, ingate=True):
    """GRUCellEx.forward, learning/modules.py:224-251."""
    dt = inp.dtype
    if ingate:
        inp = torch.sigmoid(hidden @ P[pfx + '.ig.weight'].to(dt).t() + P[pfx + '.ig.bias'].to(dt)) * inp   # :225-226
    gi = inp @ P[pfx + '.weight_ih'].to(dt).t()                   # :239
    gh = hidden @ P[pfx + '.weight_hh'].to(dt).t()                # :240
    if layernorm:
        gi, gh = _row_norm(gi), _row_norm(gh)                     # :241
    i_r, i_i, i_n = gi.chunk(3, 1)
    h_r, h_i, h_n = gh.chunk(3, 1)
    bih_r, bih_i, bih_n = P[pfx + '.bias_ih'].to(dt).chunk(3)
    bhh_r, bhh_i, bhh_n = P[pfx + '.bias_hh'].to(dt).chunk(3)
    resetgate = torch.sigmoid(i_r + bih_r + h_r + bhh_r)          # :247
    inputgate = torch.sigmoid(i_i + bih_i + h_i + bhh_i)          # :248
    newgate = torch.tanh(i_n + bih_n + resetgate * (h_n + bhh_n))  # :249
    return newgate + inputgate * (hidden - newgate)               # :250


def lstm_cell_ex(inp, hidden, P, pfx='ecc.0._cell', layernorm=True, ingate=True):
    """LSTMCellEx.forward, learning/modules.py:280-309."""
    hx, cx = hidden
    dt = inp.dtype
    if ingate:
        inp = torch.sigmoid(hx @ P[pfx + '.ig.weight'].to(dt).t() + P[pfx + '.ig.bias'].to(dt)) * inp
    gi = inp @ P[pfx + '.weight_ih'].to(dt).t() + P[pfx + '.bias_ih'].to(dt)       # :296
    gh = hx @ P[pfx + '.weight_hh'].to(dt).t() + P[pfx + '.bias_hh'].to(dt)       # :297
    if layernorm:
        gi, gh = _row_norm(gi), _row_norm(gh)
    ig_, fg, cg, og = (gi + gh).chunk(4, 1)
    cy = torch.sigmoid(fg) * cx + torch.sigmoid(ig_) * torch.tanh(cg)
    hy = torch.sigmoid(og) * torch.tanh(cy)
    return hy, cy


def rnn_graph_conv(hx, edgefeats, idxn, degs, spec: ModelSpec, rs: RnnEccSpec, P, training: bool,
                   stats=None, pfx='ecc.0', idxe=None, dec=None, rec=None):
    """RNNGraphConvModule.forward (use_pyg=0), learning/modules.py:152-183."""
    nc = hx.shape[1]
    weights = fnet_forward(edgefeats.to(hx.dtype), spec, nc if rs.vv else nc * nc, P, training, stats, pfx + '._fnet', dec, rec)  # :160
    if weights.shape[1] != nc:
        weights = weights.view(-1, nc, nc)                        # :163-164
    hxs = [hx]
    cx = torch.zeros_like(hx) if rs.kind == 'lstm' else None      # :168-169
    for _ in range(rs.nrepeats):                                  # :171
        inp = EccFunction.apply(hx, weights, nc, nc, idxn, idxe, degs, None, 1e20)   # :175
        if rs.kind == 'lstm':
            hx, cx = lstm_cell_ex(inp, (hx, cx), P, pfx + '._cell', rs.layernorm, rs.ingate)
        else:
            hx = gru_cell_ex(inp, hx, P, pfx + '._cell', rs.layernorm, rs.ingate)   # :180
        hxs.append(hx)
    return torch.cat(hxs, 1) if rs.cat_all else hx                # :183


def graph_network_forward(x, edgefeats, idxn, degs, spec: ModelSpec, P, training: bool, stats=None, dec=None, rec=None):
    """GraphNetwork.forward, learning/graphnet.py:95-98 (f / gru / lstm / r tokens)."""
    nfeat = spec.ptn_widths[1][-1]
    for d, kind, payload in parse_model_config(spec.model_config, nfeat):
        if kind == 'f':
            x = _lin(x, P, f'ecc.{d}')
        elif kind in ('gru', 'lstm'):
            x = rnn_graph_conv(x, edgefeats, idxn, degs, spec, payload[1], P, training, stats, f'ecc.{d}', dec=dec, rec=rec)
        elif kind == 'r':
            x = torch.relu(x)
        elif kind == 'b':
            x = batch_norm(x, f'ecc.{d}', P, training, stats)
        else:
            raise NotImplementedError(kind)
    return x


def weighted_cross_entropy(logits, target, class_weights=None):
    """nn.functional.cross_entropy(outputs, label_mode, weight=class_weights), ignore_index=-100
    (learning/main.py:205): sum_i w[y_i] * nll_i / sum_i w[y_i] over labelled rows."""
    valid = target != -100
    lp = torch.log_softmax(logits, 1)
    t = target.clamp(min=0)
    nll = -lp.gather(1, t.unsqueeze(1)).squeeze(1)
    w = torch.ones(logits.shape[1], dtype=logits.dtype) if class_weights is None else class_weights.to(logits.dtype)
    wi = w[t] * valid.to(logits.dtype)
    return (wi * nll).sum() / wi.sum()


# --------------------------------------------------------------------------------------
# whole step  (learning/main.py:199-208 window)
# --------------------------------------------------------------------------------------
PARAM_SUFFIXES = ('weight', 'bias', 'weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')


def is_param_key(k: str) -> bool:
    return k.split('.')[-1] in PARAM_SUFFIXES


def model_forward(batch: dict, spec: ModelSpec, P, training: bool, stats=None, dtype=torch.float32, dec=None, rec=None):
    """embeddings = CloudEmbedder.run(...); outputs = model.ecc(embeddings)  (learning/main.py:202-203)."""
    emb, idx_valid = cloud_embed(batch['clouds_flag'], batch['clouds'].to(dtype), batch['clouds_global'].to(dtype),
                                 spec, P, training, stats, dec, rec)
    logits = graph_network_forward(emb, batch['edgefeats'], batch['idxn'], batch['degs'], spec, P, training, stats, dec, rec)
    return emb, logits


def train_step(batch: dict, spec: ModelSpec, state: Dict[str, torch.Tensor], class_weights=None,
               dtype=torch.float32, update_running_stats: bool = True, monger: bool = True, dec=None, rec=None):
    """One training-step window: forward, weighted CE, backward for every parameter
    (learning/main.py:199-208).  Returns loss, logits, embeddings, {key: grad}.  `state` uses
    reference state_dict keys; running stats in `state` are updated in place like the reference
    (twice for PointNet BNs when `monger`, learning/pointnet.py:160-176)."""
    P = {}
    leaves = {}
    for k, v in state.items():
        if is_param_key(k) and v.is_floating_point():
            t = v.detach().to(dtype).clone().requires_grad_(True)
            leaves[k] = t
            P[k] = t
        else:
            P[k] = v
    stats = {}
    emb, logits = model_forward(batch, spec, P, True, stats, dtype, dec, rec)      # dec / rec: PointNet decision hooks (see _relu_dec)
    loss = weighted_cross_entropy(logits, batch['label_mode'], class_weights)
    keys = [k for k in leaves]
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    gdict = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(keys, grads)}
    if update_running_stats:
        ptn = {k: s for k, s in stats.items() if k.startswith('ptn.')}
        ecc_ = {k: s for k, s in stats.items() if not k.startswith('ptn.')}
        apply_running_stats(state, ptn, 2 if monger else 1)
        apply_running_stats(state, ecc_, 1)
    return loss.detach(), logits.detach(), emb.detach(), gdict


def clamp_and_adam(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], adam_state: dict,
                   lr=1e-2, grad_clip=1.0, betas=(0.9, 0.999), eps=1e-8, wd=0.0):
    """p.grad.clamp_(-clip, clip) (learning/main.py:210-212) then torch.optim.Adam step
    (learning/main.py:433-437; defaults betas=(0.9,0.999), eps=1e-8)."""
    adam_state['step'] = adam_state.get('step', 0) + 1
    t = adam_state['step']
    for k, g in grads.items():
        g = g.clamp(-grad_clip, grad_clip) if grad_clip > 0 else g
        if wd != 0:
            g = g + wd * params[k]
        m = adam_state.setdefault('m.' + k, torch.zeros_like(g))
        v = adam_state.setdefault('v.' + k, torch.zeros_like(g))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** t
        bc2 = 1 - betas[1] ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].data.addcdiv_(m, denom, value=-lr / bc1)
