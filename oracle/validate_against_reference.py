#!/usr/bin/env python3
"""
ORACLE PINNING + GOLDEN-VECTOR GENERATOR (test infrastructure; runs only in the build container).

Imports the upstream modules read-only from /root/reference (learning/pointnet.py, graphnet.py,
modules.py, ecc/*) with a stub `igraph`, runs them on seeded inputs and
  1. checks every function of oracle/spg_oracle.py against them (asserts), and
  2. writes tests/golden/*.npz -- inputs, reference state_dict, reference outputs / gradients --
     which travel to the GPU box (where /root/reference does not exist).

Usage:  python oracle/validate_against_reference.py [--write]
The upstream repository has no golden vectors (only property tests), so "outputs of the reference
itself, run here" is the pin.  The reference's matrix-filter backward raises on torch >= 1.5
(learning/ecc/GraphConvModule.py:146); for that one function the golden gradients come from the
reference modules with GraphConvFunction replaced by the restated oracle.EccFunction, after the
restatement has been pinned by (a) the reference forward in both modes, (b) the reference backward
in vector mode, (c) fp64 gradcheck on the reference test's fixture (test_GraphConvModule.py:29-36).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('SPG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)

from oracle import spg_oracle as O            # noqa: E402
from superpoint_graph_amd import synth        # noqa: E402


# ---- minimal igraph stand-in: only what GraphConvInfo.set_batch touches (GraphConvInfo.py:48-58) ----
class _EdgeSeq:
    def __init__(self, g, idx=None):
        self.g, self.idx = g, idx

    def __getitem__(self, idx):
        return _EdgeSeq(self.g, list(idx))

    def attributes(self):
        return list(self.g.eattrs.keys())

    def get_attribute_values(self, a):
        vals = self.g.eattrs[a]
        idx = range(len(vals)) if self.idx is None else self.idx
        return [vals[i] for i in idx]


class FakeGraph:
    def __init__(self, n, edges, edge_attrs):
        self.n, self.edges, self.eattrs = n, [tuple(int(v) for v in e) for e in edges], edge_attrs
        self.es = _EdgeSeq(self)
        self.vs = list(range(n))

    def get_edgelist(self):
        return self.edges

    def vcount(self):
        return self.n

    def indegree(self, vs, loops=True):
        d = [0] * self.n
        for _s, t in self.edges:
            d[t] += 1
        return d


def import_reference():
    ig = types.ModuleType('igraph')
    ig.Graph = FakeGraph
    sys.modules['igraph'] = ig
    sys.path.insert(0, REF)
    from learning import pointnet, graphnet, modules, ecc  # noqa
    return pointnet, graphnet, modules, ecc


def make_reference_model(spec: O.ModelSpec, seed, refmods, patch_ecc=False):
    """create_model, learning/main.py:414-431 (ecc first, then ptn which reseeds to 0)."""
    pointnet, graphnet, modules, ecc = refmods
    torch.manual_seed(seed)
    model = torch.nn.Module()
    nfeat = spec.ptn_widths[1][-1]
    model.ecc = graphnet.GraphNetwork(spec.model_config, nfeat, [spec.edge_feats] + list(spec.fnet_widths),
                                      spec.fnet_orthoinit, spec.fnet_llbias, spec.fnet_bnidx, 30000,
                                      use_pyg=0, cuda=0)
    model.ptn = pointnet.PointNet(list(spec.ptn_widths[0]), list(spec.ptn_widths[1]), list(spec.ptn_widths_stn[0]),
                                  list(spec.ptn_widths_stn[1]), spec.node_feats, spec.ptn_nfeat_stn,
                                  prelast_do=spec.ptn_prelast_do)
    return model


def randomize_bn_and_proj(model, seed):
    """Fresh BN layers have weight=1, bias=0 and the STN projection is zero-initialised
    (pointnet.py:52), which hides scale/shift and the whole STN gradient path.  Perturb them
    (deterministically) so the golden vectors exercise every term."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'proj' in name:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.weight[0] = -abs(m.weight[0])          # a negative BN scale: max-pool must pick the min
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=g))


def build_batch(seeds, n_sp, n_edges, spec, duplicate_points=True, n_classes=13):
    scenes = [synth.scene(s, n_sp=n, n_edges=e, n_feat=spec.node_feats, n_pts=spec.ptn_npts,
                          n_edge_feat=spec.edge_feats, minpts=40, small_frac=0.15, n_classes=n_classes) for s, n, e in zip(seeds, n_sp, n_edges)]
    if duplicate_points and len(scenes[0]['clouds']) > 0:
        c = scenes[0]['clouds']
        c[0, :, 100:] = c[0, :, :28]            # padded superpoint (spg.py:212-214): duplicated points => max-pool ties
    col = synth.collate_numpy(scenes)
    idxn, degs, edgefeats, edge_indexes = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(
        clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
        clouds_global=torch.from_numpy(col['clouds_global']), targets=torch.from_numpy(col['targets']),
        label_mode=torch.from_numpy(col['targets'][:, 0].copy()),
        idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs), edgefeats=torch.from_numpy(edgefeats),
        edge_indexes=torch.from_numpy(edge_indexes))
    return batch, col


def ref_gci(ecc, batch):
    gi = ecc.GraphConvInfo()
    gi._idxn, gi._idxe, gi._degrees, gi._degrees_gpu = batch['idxn'], None, batch['degs'], None
    gi._edgefeats, gi._edge_indexes = batch['edgefeats'], batch['edge_indexes']
    return gi


def maxrel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(name, a, b, tol):
    e = maxrel(a, b)
    print(f'  {name:58s} max|d|/max|ref| = {e:.3e}  (tol {tol:.0e})')
    assert e <= tol, name


def run_config(tag, spec, refmods, seeds, n_sp, n_edges, class_weights, write):
    pointnet, graphnet, modules, ecc = refmods
    print(f'== {tag}: {spec.model_config}')
    batch, col = build_batch(seeds, n_sp, n_edges, spec, n_classes=len(class_weights))

    # --- integer contract: set_batch vs the reference GraphConvInfo on a fake igraph ---
    graphs = [FakeGraph(n, e, {'f': list(f)}) for e, n, f in zip(col['edge_lists'], col['vcounts'], col['edge_feats'])]
    gi_ref = ecc.GraphConvInfo(graphs, lambda ea: (torch.from_numpy(np.asarray(ea['f'])), None))
    assert torch.equal(gi_ref._idxn, batch['idxn']) and torch.equal(gi_ref._degrees, batch['degs'])
    assert torch.equal(gi_ref._edge_indexes, batch['edge_indexes']) and torch.equal(gi_ref._edgefeats, batch['edgefeats'])
    print('  set_batch: idxn / degs / edge_indexes / edgefeats bit-exact vs reference GraphConvInfo')
    assert (batch['degs'] == 0).any(), 'fixture should contain a zero in-degree node'
    assert (batch['clouds_flag'] == -1).any(), 'fixture should contain an invalid superpoint'

    rs = [p for p in O.parse_model_config(spec.model_config, spec.ptn_widths[1][-1]) if p[1] in ('gru', 'lstm')][0][2][1]
    matrix = not rs.vv
    model = make_reference_model(spec, 1, refmods)
    randomize_bn_and_proj(model, 7)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = types.SimpleNamespace(cuda=0, ptn_mem_monger=1)

    # --- eval-mode forward ---
    model.eval()
    model.ecc.set_info([ref_gci(ecc, batch)], 0)
    emb_ref = pointnet.CloudEmbedder(args).run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    logits_ref = model.ecc(emb_ref)
    P = {k: v.clone() for k, v in state0.items()}
    emb_o, logits_o = O.model_forward(batch, spec, P, False)
    check('eval embeddings', emb_o, emb_ref, 2e-6)
    check('eval logits', logits_o, logits_ref, 5e-6)
    eval_out = dict(emb=emb_ref.detach().numpy(), logits=logits_ref.detach().numpy())

    # --- training step ---
    model.load_state_dict(state0)
    model.train()
    if matrix:
        ecc.GraphConvFunction = O.EccFunction           # reference matrix backward is unusable (see header)
    else:
        pass
    emb_ref = None
    embedder = pointnet.CloudEmbedder(args)
    emb_ref = embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    logits_ref = model.ecc(emb_ref)
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, batch['label_mode'], weight=class_weights)
    model.zero_grad()
    loss_ref.backward()
    embedder.bw_hook()
    grads_ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    state1 = {k: v.clone() for k, v in model.state_dict().items()}

    st = {k: v.clone() for k, v in state0.items()}
    loss_o, logits_o, emb_o, grads_o = O.train_step(batch, spec, st, class_weights)
    check('train embeddings', emb_o, emb_ref, 2e-5)
    check('train logits', logits_o, logits_ref, 5e-5)
    check('train loss', loss_o, loss_ref, 1e-5)
    worst = 0.0
    for k, g in grads_ref.items():
        e = maxrel(grads_o[k], g)
        # biases in front of a train-mode BatchNorm have an analytically zero gradient; both sides hold rounding noise
        noise = float(g.abs().max()) < 1e-6
        if not noise:
            worst = max(worst, e)
            assert e < 2e-4, (k, e)
    print(f'  train grads: worst max|d|/max|ref| over {len(grads_ref)} tensors = {worst:.3e}')
    for k in state1:
        if 'running' in k or 'num_batches' in k:
            assert maxrel(st[k].double(), state1[k].double()) < 1e-5, k
    print('  running stats / num_batches_tracked after the step match (monger double update)')

    # fp64 oracle on the same inputs (the "truth" both fp32 paths are compared with in the GPU tests)
    st64 = {k: v.clone() for k, v in state0.items()}
    loss64, logits64, emb64, grads64 = O.train_step(batch, spec, st64, class_weights, dtype=torch.float64,
                                                    update_running_stats=False)
    print(f'  fp32 reference vs fp64 oracle: emb {maxrel(emb_ref, emb64):.2e} logits {maxrel(logits_ref, logits64):.2e}')

    if write:
        out = os.path.join(ROOT, 'tests', 'golden', f'{tag}.npz')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        blob = {}
        for k, v in state0.items():
            blob['state0/' + k] = v.numpy()
        for k, v in state1.items():
            if 'running' in k or 'num_batches' in k:
                blob['state1/' + k] = v.numpy()
        for k, v in grads_ref.items():
            blob['grad/' + k] = v.numpy()
        for k in ('clouds_flag', 'clouds', 'clouds_global', 'targets', 'label_mode', 'idxn', 'degs', 'edgefeats',
                  'edge_indexes'):
            blob['batch/' + k] = batch[k].numpy()
        for i, (e, n, f) in enumerate(zip(col['edge_lists'], col['vcounts'], col['edge_feats'])):
            blob[f'graph/{i}/edges'], blob[f'graph/{i}/n'], blob[f'graph/{i}/feats'] = e, np.int64(n), f
        blob['eval/emb'], blob['eval/logits'] = eval_out['emb'], eval_out['logits']
        blob['train/emb'], blob['train/logits'] = emb_ref.detach().numpy(), logits_ref.detach().numpy()
        blob['train/loss'] = loss_ref.detach().numpy()
        blob['train/logits_fp64'], blob['train/emb_fp64'] = logits64.numpy(), emb64.numpy()
        blob['class_weights'] = class_weights.numpy()
        blob['spec/model_config'] = np.array(spec.model_config)
        blob['spec/ptn_widths0'], blob['spec/ptn_widths1'] = np.array(spec.ptn_widths[0]), np.array(spec.ptn_widths[1])
        blob['spec/ptn_widths_stn0'], blob['spec/ptn_widths_stn1'] = np.array(spec.ptn_widths_stn[0]), np.array(spec.ptn_widths_stn[1])
        blob['spec/ints'] = np.array([spec.node_feats, spec.edge_feats, spec.ptn_nfeat_stn, spec.fnet_llbias,
                                      spec.fnet_orthoinit, spec.fnet_bnidx, spec.ptn_npts])
        blob['spec/fnet_widths'] = np.array(spec.fnet_widths)
        np.savez_compressed(out, **blob)
        print(f'  wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)')


def check_ops(refmods, write):
    """Op-level pins: GraphConvFunction fwd (both modes) / bwd (vector), GRUCellEx, LSTMCellEx,
    get_edge_shards; plus the golden vectors for the op-level GPU tests."""
    pointnet, graphnet, modules, ecc = refmods
    print('== op-level checks')
    g = torch.Generator().manual_seed(3)
    # the reference unit-test fixture (test_GraphConvModule.py:29-36)
    n, e, cin, cout = 20, 50, 10, 15
    x = torch.randn(n, cin, generator=g, dtype=torch.float64)
    w = torch.randn(e, cin, cout, generator=g, dtype=torch.float64)
    idxn = torch.randint(0, n, (e,), generator=g)
    degs = torch.LongTensor([5, 0, 15, 20, 10])
    for lim in (1, 30, 1e10):
        out_ref = ecc.GraphConvFunction.apply(x, w, cin, cout, idxn, None, degs, None, lim)
        check(f'ecc fwd matrix fp64 (edge_mem_limit={lim:g})', O.ecc_forward(x, w, idxn, degs), out_ref, 1e-14)
        assert torch.equal(torch.tensor(O.get_edge_shards(degs.numpy(), lim)),
                           torch.tensor(ecc.utils.get_edge_shards(degs, lim)))
    assert float(O.ecc_forward(x, w, idxn, degs)[1].abs().max()) == 0.0
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b: O.EccFunction.apply(a, b, cin, cout, idxn, None, degs, None, 30), (xg, wg))
    idxe = torch.randint(0, 30, (e,), generator=g)
    w30 = torch.randn(30, cin, cout, generator=g, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b: O.EccFunction.apply(a, b, cin, cout, idxn, idxe, degs, None, 30), (xg, w30))
    print('  fp64 gradcheck of the restated backward (matrix, with and without idxe): ok')
    # vector mode: the reference backward runs -> direct comparison
    nc = 12
    xv = torch.randn(n, nc, generator=g, dtype=torch.float64, requires_grad=True)
    wv = torch.randn(e, nc, generator=g, dtype=torch.float64, requires_grad=True)
    go = torch.randn(5, nc, generator=g, dtype=torch.float64)
    out_ref = ecc.GraphConvFunction.apply(xv, wv, nc, nc, idxn, None, degs, None, 30)
    gx_ref, gw_ref = torch.autograd.grad(out_ref, (xv, wv), go)
    gx_o, gw_o = O.ecc_backward(xv.detach(), wv.detach(), go, idxn, degs)
    check('ecc bwd vector grad_input', gx_o, gx_ref, 1e-14)
    check('ecc bwd vector grad_weights', gw_o, gw_ref, 1e-14)

    # GRU / LSTM cells
    torch.manual_seed(5)
    cell = modules.GRUCellEx(32, 32, bias=True, layernorm=True, ingate=True)
    inp, hid = torch.randn(9, 32, generator=g), torch.randn(9, 32, generator=g)
    P = {'c.' + k: v for k, v in cell.state_dict().items()}
    check('GRUCellEx', O.gru_cell_ex(inp, hid, P, 'c'), cell(inp, hid), 2e-6)
    cell2 = modules.GRUCellEx(32, 32, bias=True, layernorm=False, ingate=False)
    P2 = {'c.' + k: v for k, v in cell2.state_dict().items()}
    check('GRUCellEx (no layernorm, no ingate)', O.gru_cell_ex(inp, hid, P2, 'c', False, False), cell2(inp, hid), 2e-6)
    lcell = modules.LSTMCellEx(32, 32, bias=True, layernorm=True, ingate=True)
    PL = {'c.' + k: v for k, v in lcell.state_dict().items()}
    cx = torch.randn(9, 32, generator=g)
    hy_r, cy_r = lcell(inp, (hid, cx))
    hy_o, cy_o = O.lstm_cell_ex(inp, (hid, cx), PL, 'c')
    check('LSTMCellEx hy', hy_o, hy_r, 2e-6)
    check('LSTMCellEx cy', cy_o, cy_r, 2e-6)

    if write:
        out = os.path.join(ROOT, 'tests', 'golden', 'ops.npz')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        gxm, gwm = O.ecc_backward(x, w, torch.ones(5, cout, dtype=torch.float64), idxn, degs)
        np.savez_compressed(
            out, ecc_x=x.numpy(), ecc_w=w.numpy(), ecc_idxn=idxn.numpy(), ecc_degs=degs.numpy(),
            ecc_out=ecc.GraphConvFunction.apply(x, w, cin, cout, idxn, None, degs, None, 30).numpy(),
            ecc_gx_ones=gxm.numpy(), ecc_gw_ones=gwm.numpy(),
            eccv_x=xv.detach().numpy(), eccv_w=wv.detach().numpy(), eccv_go=go.numpy(), eccv_out=out_ref.detach().numpy(),
            eccv_gx=gx_ref.numpy(), eccv_gw=gw_ref.numpy(),
            gru_in=inp.numpy(), gru_h=hid.numpy(), gru_out=cell(inp, hid).detach().numpy(),
            **{'gru_p/' + k: v.numpy() for k, v in cell.state_dict().items()},
            lstm_c=cx.numpy(), lstm_hy=hy_r.detach().numpy(), lstm_cy=cy_r.detach().numpy(),
            **{'lstm_p/' + k: v.numpy() for k, v in lcell.state_dict().items()})
        print(f'  wrote {out}')


def check_loader(write):
    """load_superpoint / augment_cloud (learning/spg.py:198-258) imported from the reference; h5py -> in-memory stub,
    transforms3d (absent) -> its closed forms, see oracle/spg_loader_oracle.py."""
    import random as pyrandom
    from oracle import spg_loader_oracle as L
    print('== loader: load_superpoint / augment_cloud')
    store = {}

    class _DS:
        def __init__(self, a):
            self._a = a
            self.shape = a.shape

        def __getitem__(self, k):
            return self._a[k]

    h5 = types.ModuleType('h5py')
    h5.File = lambda fname, mode='r': {k: _DS(v) for k, v in store[fname].items()}
    t3 = types.ModuleType('transforms3d')
    t3.zooms = types.SimpleNamespace(zfdir2mat=lambda s, d=None: L.zoom(s) if d is None else (
        np.eye(3) + (s - 1) * np.outer(d, d) / np.dot(d, d)))
    t3.axangles = types.SimpleNamespace(axangle2mat=lambda ax, a: L.rot_z(a) if list(ax) == [0, 0, 1] else None)
    sk = sys.modules.get('sklearn')
    sys.modules['h5py'], sys.modules['transforms3d'] = h5, t3
    from learning import spg as refspg

    rng = np.random.default_rng(77)
    counts = [5, 39, 40, 127, 128, 129, 400, 1, 3000, 64]
    pts = [np.concatenate([rng.normal(size=(n, 3)) * rng.uniform(0.2, 5) + rng.normal(size=3) * 10,
                           rng.uniform(-0.5, 0.5, size=(n, 8)), rng.uniform(0, 1, size=(n, 3))], 1).astype(np.float32)
           for n in counts]
    pts[4][:, :3] = pts[4][0, :3]                       # a degenerate superpoint: all points identical (diameter 0)
    store['scene.h5'] = {str(i): p for i, p in enumerate(pts)}
    points = np.concatenate(pts, 0)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ids = np.arange(len(counts))
    out = {}
    for tag, attribs, norm in (('s3dis', 'xyzrgbelpsvXYZ', 1), ('sema3d', 'xyzrgbelpsv', 1), ('nonorm', 'xyzelpsv', 0)):
        args = types.SimpleNamespace(ptn_minpts=40, ptn_npts=128, pc_xyznormalize=norm, pc_attribs=attribs,
                                     pc_augm_scale=0, pc_augm_rot=0, pc_augm_mirror_prob=0, pc_augm_jitter=0)
        mine = L.load_batch(points, offsets, ids, 40, 128, norm, attribs, train=False, test_seed_offset=3)
        k = 0
        for i in range(len(counts)):
            P, d = refspg.load_superpoint(args, 'scene.h5', i, False, 3)
            if P is None:
                assert mine['flag'][i] == -1 and d == counts[i]
                continue
            assert mine['flag'][i] == 0 and mine['slot'][i] == k
            assert np.array_equal(P.T, mine['clouds'][k]) and np.array_equal(d, mine['diam'][k:k + 1]), (tag, i)
            k += 1
        print(f'  test-mode {tag:7s}: {k} clouds bit-equal to the reference (flags, diameters, resampling stream)')
        out[tag] = mine
    # training mode: global numpy / python random streams, full augmentation
    args = types.SimpleNamespace(ptn_minpts=40, ptn_npts=128, pc_xyznormalize=1, pc_attribs='xyzrgbelpsvXYZ',
                                 pc_augm_scale=1.1, pc_augm_rot=1, pc_augm_mirror_prob=1.0, pc_augm_jitter=1)
    np.random.seed(5); pyrandom.seed(6)
    ref = [refspg.load_superpoint(args, 'scene.h5', i, True, 0) for i in range(len(counts))]
    np.random.seed(5); pyrandom.seed(6)
    mine = L.load_batch(points, offsets, ids, 40, 128, 1, 'xyzrgbelpsvXYZ', train=True,
                        augm=dict(scale=1.1, rot=1, mirror_prob=1.0, jitter=1), nprandom=np.random, pyrandom=pyrandom)
    k = 0
    for i, (P, d) in enumerate(ref):
        if P is None:
            continue
        assert np.array_equal(P.T, mine['clouds'][k]) and np.array_equal(d, mine['diam'][k:k + 1]), i
        k += 1
    print(f'  train-mode (scale+rot+mirror+jitter): {k} clouds bit-equal to the reference')
    out['train'] = mine
    if write:
        path = os.path.join(ROOT, 'tests', 'golden', 'loader.npz')
        arrs = dict(points=points, offsets=offsets, ids=ids)
        for tag, m in out.items():
            for key in ('flag', 'slot', 'sample_idx', 'clouds', 'diam', 'M'):
                arrs[f'{tag}/{key}'] = m[key]
        arrs['train/noise'] = out['train']['noise']
        np.savez_compressed(path, **arrs)
        print(f'  wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')
    del sys.modules['h5py'], sys.modules['transforms3d']


def check_metrics(write):
    """metrics.ConfusionMatrix (learning/metrics.py) imported from the reference, driven like eval_final does."""
    from oracle import spg_metrics_oracle as MO
    from learning import metrics as refmetrics
    print('== metrics: ConfusionMatrix / eval accounting')
    rng = np.random.default_rng(5)
    N, C, S = 300, 13, 3
    samples = [rng.normal(size=(N, C)).astype(np.float32) for _ in range(S)]
    samples[1][7] = samples[0][7] = samples[2][7] = 0.25          # an exact tie: the first arg-max wins
    label_vec = rng.integers(0, 4000, size=(N, C)).astype(np.int64) * (rng.random((N, C)) < 0.3)
    label_mode = np.where(label_vec.sum(1) == 0, -100, label_vec.argmax(1)).astype(np.int64)
    label_vec[:, 11] = 0                                          # a class that never occurs as ground truth
    out = {}
    for tag, smp in (('multi', samples), ('single', samples[:1])):
        ref = refmetrics.ConfusionMatrix(C)
        o = np.mean(np.stack(smp, 0), 0) if len(smp) > 1 else smp[0]
        idx = label_mode != -100
        ref.count_predicted_batch(label_vec[idx], np.argmax(o[idx], 1))
        pred, cm, correct, counted = MO.aggregate(smp, label_mode, label_vec, C)
        assert np.array_equal(cm, ref.confusion_matrix) and np.array_equal(pred, np.argmax(o, 1))
        iou, oa, miou, mca = MO.scores(cm)
        assert iou == ref.get_intersection_union_per_class() and oa == ref.get_overall_accuracy()
        assert miou == ref.get_average_intersection_union() and mca == ref.get_mean_class_accuracy()
        print(f'  {tag}: confusion matrix, predictions and all four scores equal the reference bit for bit '
              f'(OA {oa:.4f}, mIoU {miou:.4f}, {counted} counted superpoints)')
        out[tag] = dict(pred=pred, cm=cm, correct=np.int64(correct), counted=np.int64(counted),
                        iou=np.array(iou), oa=np.float64(oa), miou=np.float64(miou), mca=np.float64(mca))
    if write:
        path = os.path.join(ROOT, 'tests', 'golden', 'metrics.npz')
        arrs = dict(samples=np.stack(samples), label_vec=label_vec, label_mode=label_mode)
        for tag, d in out.items():
            for k, v in d.items():
                arrs[f'{tag}/{k}'] = v
        np.savez_compressed(path, **arrs)
        print(f'  wrote {path}')


LOCAL_SPEC = dict(node_feats=6, ptn_nfeat_stn=0, ptn_widths=((32, 128), (34, 32, 32, 4)), ptn_widths_stn=((16, 64), (32, 16)))


def make_local_model(pointnet_mod, seed=3):
    """create_model of supervized_partition/supervized_partition.py:411-421 with its default flags (ptn_widths [[32,128],[34,32,32,4]],
    ptn_widths_stn [[16,64],[32,16]], ptn_nfeat_stn 2, stn_as_global 1, global_feat 'eXYrgb' -> 6 + 4 + 1 = 11 global features,
    xyz + rgb = 6 point features); works for the reference's module and for the product's."""
    torch.manual_seed(seed)
    model = torch.nn.Module()
    model.stn = pointnet_mod.STNkD(2, [16, 64], [32, 16])
    model.ptn = pointnet_mod.PointNet([32, 128], [34, 32, 32, 4], [], [], 6, 0, prelast_do=0, nfeat_global=11, is_res=False, last_bn=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():             # the zero-initialised projection would hide the transform path; non-trivial BatchNorm
        model.stn.proj.weight.copy_(0.05 * torch.randn(model.stn.proj.weight.shape, generator=g))
        model.stn.proj.bias.copy_(0.05 * torch.randn(model.stn.proj.bias.shape, generator=g))
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=g))
    return model


def local_inputs(n, k=20, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 6, k, generator=g), torch.randn(n, 7, generator=g), torch.randn(n, 4, generator=g)


def check_local_embedder(refmods, write):
    """LocalCloudEmbedder.run_batch (learning/pointnet.py:182-207) of the IMPORTED reference vs oracle.local_cloud_embed:
    train-mode forward + all gradients + running statistics at 700 clouds (golden for the GPU test), eval-mode forward, and
    the chunk boundary (2^16 - 1 clouds per BatchNorm batch in training mode) at 2^16 + 40 clouds."""
    pointnet = refmods[0]
    print('== LocalCloudEmbedder (supervised partition embedder)')
    spec = O.ModelSpec(**LOCAL_SPEC)
    args = types.SimpleNamespace(ptn_nfeat_stn=2, stn_as_global=1)
    blob = {}
    for tag, n in (('n700', 700), ('chunk', 2 ** 16 + 40)):
        model = make_local_model(pointnet)
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        clouds, cg, w = local_inputs(n)
        model.train()
        emb = pointnet.LocalCloudEmbedder(args).run_batch(model, clouds, cg)
        (emb * w).sum().backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        state1 = {k: v.clone() for k, v in model.state_dict().items()}
        # oracle, same state
        P, leaves = {}, {}
        for k, v in state0.items():
            P[k] = v.clone().requires_grad_(True) if O.is_param_key(k) and v.is_floating_point() else v.clone()
            if P[k].requires_grad:
                leaves[k] = P[k]
        emb_o = O.local_cloud_embed(clouds, cg, spec, P, True, 2, True, update_running=True)
        go = torch.autograd.grad((emb_o * w).sum(), list(leaves.values()), allow_unused=True)
        check(f'{tag}: train embeddings', emb_o, emb, 2e-5)
        worst = 0.0
        for (k, _), g in zip(leaves.items(), go):
            # biases in front of a train-mode BatchNorm: analytically zero, round-off on both sides
            if float(grads[k].abs().max()) > 1e-5 * max(float(v.abs().max()) for v in grads.values()):
                worst = max(worst, maxrel(g, grads[k]))
        print(f'  {tag}: gradients oracle vs reference worst {worst:.2e}')
        assert worst < (2e-4 if n < 1000 else 2e-3)
        for k in state1:
            if 'running' in k:
                assert maxrel(P[k].double(), state1[k].double()) < 1e-5, k
            if 'num_batches' in k:
                assert int(P[k]) == int(state1[k]), k
        print(f'  {tag}: running statistics / num_batches_tracked match (one BatchNorm batch per chunk of {O.LOCAL_CHUNK})')
        model.load_state_dict(state0)
        model.eval()
        with torch.no_grad():
            emb_e = pointnet.LocalCloudEmbedder(args).run_batch(model, clouds, cg)
        emb_eo = O.local_cloud_embed(clouds, cg, spec, {k: v.clone() for k, v in state0.items()}, False, 2, True)
        check(f'{tag}: eval embeddings', emb_eo, emb_e, 5e-6)
        if tag == 'n700':
            blob.update({'n700/train_emb': emb.detach().numpy(), 'n700/eval_emb': emb_e.numpy(), 'state0_sha256': np.array(state_digest(state0))})
            blob.update({'state0/' + k: v.numpy() for k, v in state0.items()})
            for k, v in grads.items():
                blob['n700/grad/' + k] = v.numpy()
            for k, v in state1.items():
                if 'running' in k:
                    blob['n700/state1/' + k] = v.numpy()
        else:         # the large case travels as a few rows + checksums (the full [65576, 4] output is regenerable, not needed)
            rows = torch.tensor([0, 1, 65534, 65535, 65536, n - 1])
            blob.update({'chunk/n': np.int64(n), 'chunk/rows': rows.numpy(), 'chunk/train_emb_rows': emb.detach()[rows].numpy(),
                         'chunk/eval_emb_rows': emb_e[rows].numpy(), 'chunk/train_emb_colsum': emb.detach().double().sum(0).numpy(),
                         'chunk/eval_emb_colsum': emb_e.double().sum(0).numpy()})
            for k, v in state1.items():
                if 'running' in k:
                    blob['chunk/state1/' + k] = v.numpy()
            # float64 referee for the gradients at this size: with 1.3 M points / 65 535 clouds per BatchNorm batch the
            # reference's own fp32 run is 7e-4 .. 3e-3 from float64 on the tensors in front of ReLU near-ties
            P64, lv64 = {}, {}
            for k, v in state0.items():
                P64[k] = v.double() if v.is_floating_point() else v.clone()
                if O.is_param_key(k) and v.is_floating_point():
                    P64[k] = P64[k].requires_grad_(True)
                    lv64[k] = P64[k]
            e64 = O.local_cloud_embed(clouds.double(), cg.double(), spec, P64, True, 2, True)
            g64 = dict(zip(lv64, torch.autograd.grad((e64 * w.double()).sum(), list(lv64.values()), allow_unused=True)))
            for k in ('stn.convs.0.weight', 'ptn.convs.0.weight', 'ptn.fcs.0.weight', 'ptn.fcs.9.weight'):
                blob['chunk/grad/' + k] = grads[k].numpy()
                blob['chunk/grad64/' + k] = g64[k].numpy()
                print(f'  chunk: {k}: reference fp32 vs float64 oracle {maxrel(grads[k], g64[k]):.2e}')
    if write:
        out = os.path.join(ROOT, 'tests', 'golden', 'local_embedder.npz')
        np.savez_compressed(out, **blob)
        print(f'  wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)')


def state_digest(state):
    """sha256 over the float tensors of a state_dict in key order (integrity check of a stored initial state).  The state
    itself has to travel with the golden: regenerating it from the seeds is NOT portable -- torch's CPU initialisers
    (vectorised normal_, LAPACK QR of the orthogonal init) give different bits on the GPU box's EPYC than on the build
    container's Xeon (measured, round 3)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(state):
        v = state[k]
        if v.is_floating_point():
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def baseline_size_batch(seed=0):
    """The BASELINE unit scene (1000 superpoints x 128 points x 14 features, 5000 superedges) as a batch."""
    col = synth.collate_numpy([synth.scene(seed, n_sp=1000, n_edges=5000)])
    idxn, degs, ef, ei = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    return dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                edgefeats=torch.from_numpy(ef), edge_indexes=torch.from_numpy(ei),
                label_mode=torch.from_numpy(col['targets'][:, 0].copy()))


def check_baseline_size(refmods, write):
    """The IMPORTED reference on the BASELINE-size scene itself (S3DIS production model): eval forward, train forward, loss
    and all gradients -- the headline configuration's parity is then a direct comparison with the reference instead of
    reference -> oracle (small fixtures) -> oracle (large) -> HIP.  Stored: the initial state, outputs and gradients."""
    pointnet, graphnet, modules, ecc = refmods
    print('== BASELINE-size scene (1000 superpoints x 128 pts, 5000 superedges), gru_10_0,f_13, imported reference')
    spec = O.ModelSpec()
    batch = baseline_size_batch(0)
    model = make_reference_model(spec, 1, refmods)
    randomize_bn_and_proj(model, 7)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = types.SimpleNamespace(cuda=0, ptn_mem_monger=1)
    cw = torch.linspace(0.5, 1.5, 13)
    model.eval()
    model.ecc.set_info([ref_gci(ecc, batch)], 0)
    with torch.no_grad():
        emb_e = pointnet.CloudEmbedder(args).run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
        logits_e = model.ecc(emb_e)
    emb_o, logits_o = O.model_forward(batch, spec, {k: v.clone() for k, v in state0.items()}, False)
    check('eval embeddings (oracle vs reference)', emb_o, emb_e, 5e-6)
    check('eval logits (oracle vs reference)', logits_o, logits_e, 1e-5)
    model.load_state_dict(state0)
    model.train()
    ecc.GraphConvFunction = O.EccFunction              # matrix-filter backward: restated (header)
    embedder = pointnet.CloudEmbedder(args)
    emb_t = embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    logits_t = model.ecc(emb_t)
    loss_t = torch.nn.functional.cross_entropy(logits_t, batch['label_mode'], weight=cw)
    model.zero_grad()
    loss_t.backward()
    embedder.bw_hook()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    state1 = {k: v.clone() for k, v in model.state_dict().items()}
    st = {k: v.clone() for k, v in state0.items()}
    loss_o, logits_o, emb_o, grads_o = O.train_step(batch, spec, st, cw)
    check('train embeddings (oracle vs reference)', emb_o, emb_t, 2e-5)
    check('train logits (oracle vs reference)', logits_o, logits_t, 5e-5)
    check('train loss (oracle vs reference)', loss_o, loss_t, 1e-5)
    after_pool = [k for k in grads if (k.startswith('ecc.') or k.startswith('ptn.fcs.')) and float(grads[k].abs().max()) > 1e-6]
    worst = max(maxrel(grads_o[k], grads[k]) for k in after_pool)
    print(f'  gradients behind the max-pool (ECC, FC head): oracle vs reference worst {worst:.2e}')
    assert worst < 2e-4
    if write:
        out = os.path.join(ROOT, 'tests', 'golden', 'baseline_size.npz')
        blob = {'state0_sha256': np.array(state_digest(state0)), 'class_weights': cw.numpy(),
                **{'state0/' + k: v.numpy() for k, v in state0.items()},
                'eval/emb': emb_e.numpy(), 'eval/logits': logits_e.numpy(),
                'train/emb': emb_t.detach().numpy(), 'train/logits': logits_t.detach().numpy(), 'train/loss': loss_t.detach().numpy()}
        for k, v in grads.items():
            blob['grad/' + k] = v.numpy()
        for k, v in state1.items():
            if 'running' in k:
                blob['state1/' + k] = v.numpy()
        np.savez_compressed(out, **blob)
        print(f'  wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)')


def two_scene_batch(seeds=(0, 1)):
    """The reference's default --batch_size 2 (learning/main.py:49): two BASELINE-shaped scenes in one batch (2000 superpoints,
    10 000 superedges)."""
    col = synth.collate_numpy([synth.scene(s, n_sp=1000, n_edges=5000) for s in seeds])
    idxn, degs, ef, ei = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    return dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                edgefeats=torch.from_numpy(ef), edge_indexes=torch.from_numpy(ei),
                label_mode=torch.from_numpy(col['targets'][:, 0].copy()))


def check_two_scenes(refmods, write):
    """The IMPORTED reference on a 2-scene batch -- its own default batch size, BASELINE.json configs[2] -- with the initial
    state of the BASELINE-size golden (same seeds; only its digest is stored here): train forward, loss, all gradients,
    running statistics.  Multi-scene batches take other launch paths than the unit scene (2000 nodes per RNN-ECC launch,
    BatchNorm statistics over both scenes)."""
    pointnet, graphnet, modules, ecc = refmods
    print('== 2-scene batch (2 x 1000 superpoints, 10 000 superedges), gru_10_0,f_13, imported reference')
    spec = O.ModelSpec()
    batch = two_scene_batch()
    model = make_reference_model(spec, 1, refmods)
    randomize_bn_and_proj(model, 7)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = types.SimpleNamespace(cuda=0, ptn_mem_monger=1)
    cw = torch.linspace(0.5, 1.5, 13)
    model.train()
    ecc.GraphConvFunction = O.EccFunction              # matrix-filter backward: restated (header)
    model.ecc.set_info([ref_gci(ecc, batch)], 0)
    embedder = pointnet.CloudEmbedder(args)
    emb_t = embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    logits_t = model.ecc(emb_t)
    loss_t = torch.nn.functional.cross_entropy(logits_t, batch['label_mode'], weight=cw)
    model.zero_grad()
    loss_t.backward()
    embedder.bw_hook()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    state1 = {k: v.clone() for k, v in model.state_dict().items()}
    st = {k: v.clone() for k, v in state0.items()}
    loss_o, logits_o, emb_o, grads_o = O.train_step(batch, spec, st, cw)
    check('train embeddings (oracle vs reference)', emb_o, emb_t, 2e-5)
    check('train logits (oracle vs reference)', logits_o, logits_t, 5e-5)
    check('train loss (oracle vs reference)', loss_o, loss_t, 1e-5)
    after_pool = [k for k in grads if (k.startswith('ecc.') or k.startswith('ptn.fcs.')) and float(grads[k].abs().max()) > 1e-6]
    worst = max(maxrel(grads_o[k], grads[k]) for k in after_pool)
    print(f'  gradients behind the max-pool (ECC, FC head): oracle vs reference worst {worst:.2e}')
    assert worst < 2e-4
    if write:
        out = os.path.join(ROOT, 'tests', 'golden', 'two_scenes.npz')
        blob = {'state0_sha256': np.array(state_digest(state0)), 'class_weights': cw.numpy(),
                'train/emb': emb_t.detach().numpy(), 'train/logits': logits_t.detach().numpy(), 'train/loss': loss_t.detach().numpy()}
        for k, v in grads.items():
            blob['grad/' + k] = v.numpy()
        for k, v in state1.items():
            if 'running' in k:
                blob['state1/' + k] = v.numpy()
        np.savez_compressed(out, **blob)
        print(f'  wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB)')


def check_sp_graph(write):
    """partition/graphs.py:75-210 compute_sp_graph: the IMPORTED reference function (scipy's `Delaunay.vertices` is today's
    `.simplices`: the attribute is supplied, nothing else is touched) against oracle/spg_partition_oracle.py on a synthetic
    labelled cloud that exercises every branch (1-, 2- and many-point components, duplicated points, with / without labels and
    d_max); writes tests/golden/sp_graph.npz (inputs incl. the tetrahedra, reference outputs)."""
    import importlib
    import warnings
    import scipy.spatial
    from oracle import spg_partition_oracle as P
    sys.path.insert(0, os.path.join(REF, 'partition'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        graphs = importlib.import_module('graphs')

    class _Delaunay(scipy.spatial.Delaunay):
        @property
        def vertices(self):
            return self.simplices
    graphs.Delaunay = _Delaunay
    blob = {}
    for tag, seed, d_max, with_labels in (('a', 0, 1.5, True), ('b', 1, 0.0, True), ('c', 2, 0.8, False)):
        xyz, comp, components, labels = P.synthetic_cloud(seed)
        n_labels = 5 if with_labels else 0
        lab = labels if with_labels else np.zeros(0, dtype=np.int64)
        ref = graphs.compute_sp_graph(xyz, d_max, comp, components, lab, n_labels)
        tets = scipy.spatial.Delaunay(xyz).simplices
        mine = P.sp_graph_after_triangulation(xyz, d_max, comp, components, lab, n_labels, tets)
        worst = 0.0
        for k, a in ref.items():
            b = mine[k]
            if isinstance(a, (bool, list)):
                assert (a == b) if isinstance(a, bool) else len(b) == 0, k
                continue
            assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape)
            if a.dtype.kind in 'ui' or k.startswith('sp_') or 'ratio' in k or k == 'se_delta_centroid':
                assert np.array_equal(a, b), f'sp_graph {tag}: {k} differs'      # integers and everything order-independent: bit-equal
            else:                                                                # sums over a group whose edge order the reference leaves open
                worst = max(worst, float(np.abs(a.astype(np.float64) - b).max() / np.abs(a).max()))
        assert worst < 1e-6, worst
        print(f'  sp_graph {tag}: {len(xyz)} points, {len(components)} components, {len(ref["source"])} superedges: integers / superpoint '
              f'features bit-equal, offset statistics {worst:.1e}')
        blob.update({f'{tag}/xyz': xyz, f'{tag}/comp': comp, f'{tag}/labels': lab, f'{tag}/tets': tets.astype(np.int32),
                     f'{tag}/d_max': np.float64(d_max), f'{tag}/n_labels': np.int64(n_labels)})
        for k, a in ref.items():
            if not isinstance(a, (bool, list)):
                blob[f'{tag}/ref/{k}'] = a
    if write:
        out = os.path.join(ROOT, 'tests', 'golden', 'sp_graph.npz')
        np.savez_compressed(out, **blob)
        print('  wrote', out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--write', action='store_true', help='write tests/golden/*.npz')
    ap.add_argument('--only', default='', help='run (and write) only this model fixture tag')
    a = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    refmods = import_reference()
    if not a.only:
        check_ops(refmods, a.write)
    if a.only in ('', 'loader'):
        check_loader(a.write)
    if a.only in ('', 'metrics'):
        check_metrics(a.write)
    if a.only in ('', 'baseline_size'):
        check_baseline_size(refmods, a.write)
    if a.only in ('', 'two_scenes'):
        check_two_scenes(refmods, a.write)
    if a.only in ('', 'sp_graph'):
        check_sp_graph(a.write)
    if a.only in ('', 'local_embedder'):
        check_local_embedder(refmods, a.write)
    cw = torch.linspace(0.5, 1.5, 13)
    # S3DIS production config (S3DIS.md:26-28): matrix filters, 10 GRU iterations, state concat
    if a.only in ('', 's3dis_gru10_matrix'):
        run_config('s3dis_gru10_matrix', O.ModelSpec(), refmods, seeds=(11, 12), n_sp=(30, 19), n_edges=(96, 50),
                   class_weights=cw, write=a.write)
    # vector filters, small PointNet, no concat (vKITTI-style widths, vKITTI3D.md:47-51; 11 features as Semantic3D)
    spec_v = O.ModelSpec(model_config='gru_4_1_1_1_0,f_8', node_feats=11, ptn_nfeat_stn=11,
                         ptn_widths=((64, 64, 128), (64, 32, 32)), ptn_widths_stn=((32, 64), (32, 16)))
    if a.only in ('', 'vector_gru4_small'):
        run_config('vector_gru4_small', spec_v, refmods, seeds=(21,), n_sp=(40,), n_edges=(130,),
                   class_weights=torch.ones(8), write=a.write)
    # LSTMCellEx variant of the recurrent module (modules.py:262-316, `lstm_R` token of graphnet.py:52-70):
    # matrix filters, 3 iterations, state concat
    spec_l = O.ModelSpec(model_config='lstm_3_0,f_8', node_feats=11, ptn_nfeat_stn=11,
                         ptn_widths=((64, 64, 128), (64, 32, 32)), ptn_widths_stn=((32, 64), (32, 16)))
    if a.only in ('', 'lstm3_matrix_small'):
        run_config('lstm3_matrix_small', spec_l, refmods, seeds=(36,), n_sp=(36,), n_edges=(120,),
                   class_weights=torch.ones(8), write=a.write)
    print('ALL ORACLE CHECKS PASSED')


if __name__ == '__main__':
    main()
