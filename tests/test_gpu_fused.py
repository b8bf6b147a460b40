"""spg_train_step (superpoint_graph_amd/fused.py: FusedStep; include/spg_hip.h): forward + backward of a training step as ONE
C call, in which the filter network's forward and the tail of the RNN-ECC backward travel as riders next to PointNet's
launches.  Same kernels and arithmetic as the module-level path (CloudEmbedder.run -> model.ecc -> cross_entropy -> backward
-> bw_hook), so with the classifier / cross entropy as separate launches (spg_tune key 15 = 1) EVERYTHING must be bit-identical:
loss, logits, descriptors, all gradients, the BatchNorm running statistics and batch counters -- over several optimiser steps,
with class weights / sum reduction, with too-small superpoints, on the golden fixtures (matrix and vector filters, LSTM cell)
and at BASELINE size.  The default step computes classifier + cross entropy inside the one-launch GRU recurrence (SpgEccHead,
csrc/spg_ecc.h: sequential fma chains instead of MFMA chunks): everything in front of the classifier stays identical, the rest
is compared at fp32 round-off on the first step (same parameters on both sides) and must be deterministic."""
import types

import numpy as np
import pytest
import torch

from conftest import build_model, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _modular(model, arena, batch, cw, reduction, monger=True):
    from superpoint_graph_amd import ops
    from superpoint_graph_amd.learning import ecc, pointnet
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=int(monger)))
    arena.zero_grad()
    emb = embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    logits = model.ecc(emb)
    loss = ops.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw, reduction=reduction)
    loss.backward(arena.one)
    embedder.bw_hook()
    return loss.detach().clone(), logits.detach().clone(), emb.detach().clone()


def _fused(model, arena, step, batch):
    from superpoint_graph_amd.learning import ecc
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    loss, logits = step(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi, batch['label_mode'].to(DEV))
    return loss.clone(), logits.clone(), step.embeddings.clone()


def _run(spec, state0, batch, cw, reduction, nsteps, monger, fused):
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep, supports
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    assert supports(model)
    step = FusedStep(model, arena, class_weights=cw, reduction=reduction, ptn_mem_monger=monger) if fused else None
    rec = []
    for it in range(nsteps):
        r = _fused(model, arena, step, batch) if fused else _modular(model, arena, batch, cw, reduction, monger)
        torch.cuda.synchronize()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        arena.adam_step(lr=1e-3, grad_clip=1.0)
        rec.append((r, grads))
    return rec, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _assert_identical(a, b, nsteps, what=''):
    (ra, sa), (rb, sb) = a, b
    for it in range(nsteps):
        (la, lga, ea), ga = ra[it]
        (lb, lgb, eb), gb = rb[it]
        assert torch.equal(la, lb), (what, it, float(la), float(lb))
        assert torch.equal(lga, lgb) and torch.equal(ea, eb), (what, it)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (what, it, k)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), (what, k)


def _compare(spec, state0, batch, cw, reduction, nsteps=3, monger=True, head=True):
    """head: the model is one the in-recurrence classifier / cross entropy serves (GRU, <= 2048 nodes per round, cat_all = 0)."""
    from conftest import maxrel, noise_grad
    from superpoint_graph_amd import _lib
    hip = _lib.lib()
    modules = _run(spec, state0, batch, cw, reduction, nsteps, monger, False)
    old = hip.spg_tune(15, 1)
    try:
        separate = _run(spec, state0, batch, cw, reduction, nsteps, monger, True)
    finally:
        hip.spg_tune(15, old)
    _assert_identical(modules, separate, nsteps, 'modules vs one call with separate classifier / loss launches')
    assert float(modules[0][0][0][0]) > 0 and np.isfinite(float(modules[0][-1][0][0]))
    fused = _run(spec, state0, batch, cw, reduction, nsteps, monger, True)
    if not head:      # nothing for the head to take: the default step IS the step with separate launches
        _assert_identical(modules, fused, nsteps, 'modules vs one call (nothing for the head)')
        return
    again = _run(spec, state0, batch, cw, reduction, nsteps, monger, True)
    _assert_identical(fused, again, nsteps, 'one call, twice')      # deterministic
    (la, lga, ea), ga = modules[0][0]
    (lb, lgb, eb), gb = fused[0][0]
    assert torch.equal(ea, eb)                   # PointNet's forward is untouched
    assert maxrel(lb, la) < 2e-6 and maxrel(lgb, lga) < 2e-6, (float(la), float(lb), maxrel(lgb, lga))
    assert not torch.equal(lga, lgb) or lga.numel() < 64, 'the head did not run (logits bit-identical to the MFMA classifier)'
    worst = ('', 0.0)
    for k in ga:
        if noise_grad(k, ga):
            continue
        e = maxrel(gb[k], ga[k])
        worst = (k, e) if e > worst[1] else worst
    assert worst[1] < 2e-5, worst
    for it in range(1, nsteps):                  # later steps: different round-off through Adam; the loss stays close
        assert abs(float(fused[0][it][0][0]) - float(modules[0][it][0][0])) <= 2e-3 * abs(float(modules[0][it][0][0])), it


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'])
@pytest.mark.parametrize('reduction', ['mean', 'sum'])
def test_fused_step_bit_identical_to_modules_on_goldens(hip, tag, reduction):
    spec, batch, state0, g = load_golden(tag)
    cw = torch.from_numpy(g['class_weights']).to(DEV) if 'class_weights' in g.files else None
    _compare(spec, state0, batch, cw, reduction, head=(tag != 'lstm3_matrix_small'))


@pytest.mark.parametrize('n_sp,n_edges,small', [(1000, 5000, 0.0), (600, 2900, 0.1), (2000, 9000, 0.0)])
def test_fused_step_bit_identical_at_scene_size(hip, n_sp, n_edges, small):
    """BASELINE-size scene (persistent RNN-ECC), a scene with too-small superpoints (zero descriptors, B < N) and a 2000-node
    batch (persistent RNN-ECC with two workgroups per CU)."""
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    spec = O.ModelSpec()
    col = synth.collate_numpy([synth.scene(5, n_sp=n_sp, n_edges=n_edges, small_frac=small)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    if small > 0:
        assert int((batch['clouds_flag'] != 0).sum()) > 0
    torch.manual_seed(1)
    ref = build_model(spec)
    with torch.no_grad():
        ref.ptn.stn.proj.weight.normal_(0, 0.02)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    _compare(spec, state0, batch, None, 'mean', nsteps=2, monger=(n_sp != 600))


def _fused_parts(model, arena, step, batch):
    from superpoint_graph_amd.learning import ecc
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone(), parts=batch['parts'])
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    loss, logits = step(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi, batch['label_mode'].to(DEV))
    torch.cuda.synchronize()
    return loss.clone(), logits.clone(), {k: p.grad.clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize('model_config,n_scenes', [('gru_10_0,f_13', 3), ('gru_4_1_1_1_0,f_8', 5)])
def test_head_inside_the_recurrence_on_multi_scene_batches(hip, model_config, n_scenes):
    """Classifier + cross entropy inside the one-launch recurrence when the batch needs ROUNDS (3000 / 5000 nodes: rounds of
    whole scenes, csrc/spg_ecc.hip) -- cat_all classifier (352 inputs) and h^R-only classifier (32 inputs, vector filters),
    class weights, unlabelled superpoints (ignore_index): against the same call with separate classifier / loss launches
    (spg_tune key 15) at fp32 round-off, with the classifier's parameter gradients from the service workgroups and (key 6) as a
    job of a grouped launch; deterministic; persistent launches without time-outs."""
    from conftest import maxrel, noise_grad
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    spec = O.ModelSpec(model_config=model_config)
    col = synth.collate_numpy([synth.scene(11 + k, n_sp=1000, n_edges=4800 + 100 * k) for k in range(n_scenes)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    n_classes = int(model_config.rsplit('_', 1)[1])
    labels = torch.from_numpy(col['targets'][:, 0].copy()) % n_classes
    labels[::7] = -100                                   # unlabelled superpoints (learning/main.py: ignore_index)
    parts = np.concatenate([[0], np.cumsum(col['vcounts'])]).tolist()
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=labels, parts=parts)
    torch.manual_seed(3)
    ref = build_model(spec)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    cw = (0.5 + torch.rand(n_classes)).to(DEV)
    hip.spg_ecc_persistent_errors_clear()

    def run(k15, k6):
        o15, o6 = hip.spg_tune(15, k15), hip.spg_tune(6, k6)
        try:
            model = build_model(spec, state0).to(DEV).train()
            arena = FlatParameters(model, lazy_zero=True, host_counters=True)
            return _fused_parts(model, arena, FusedStep(model, arena, class_weights=cw), batch)
        finally:
            hip.spg_tune(15, o15); hip.spg_tune(6, o6)

    sep, head, head2, leaf = run(1, 0), run(0, 0), run(0, 0), run(0, 1)
    assert hip.spg_ecc_persistent_errors() == 0
    assert torch.equal(head[0], head2[0]) and torch.equal(head[1], head2[1])
    for k in head[2]:
        assert torch.equal(head[2][k], head2[2][k]), k      # deterministic
    for name, other in (('service workgroups', head), ('grouped-launch job', leaf)):
        assert maxrel(other[0], sep[0]) < 2e-6 and maxrel(other[1], sep[1]) < 2e-6, name
        assert not torch.equal(other[1], sep[1]), 'the head did not run'
        worst = ('', 0.0)
        for k in sep[2]:
            if noise_grad(k, sep[2]):
                continue
            e = maxrel(other[2][k], sep[2][k])
            worst = (k, e) if e > worst[1] else worst
        assert worst[1] < 2e-5, (name, worst)
    print(f'{model_config} x {n_scenes} scenes: loss {float(head[0]):.6f} (separate launches {float(sep[0]):.6f})')


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small', 'scene600'])
def test_infer_step_bit_identical_to_modules(hip, tag):
    """spg_infer_step (FusedStep.infer): the evaluation forward (model.eval(), running statistics) as one call -- logits
    bit-identical to model.ecc(CloudEmbedder.run(...)) under no_grad, on the goldens and on a scene with too-small superpoints
    (zero descriptors); after a training step (so that the running statistics are not the initial ones); refused in train mode."""
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    from superpoint_graph_amd.learning import ecc, pointnet
    if tag == 'scene600':
        spec = O.ModelSpec()
        col = synth.collate_numpy([synth.scene(5, n_sp=600, n_edges=2900, small_frac=0.1)])
        idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
        batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                     clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                     edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
        torch.manual_seed(2)
        state0 = {k: v.clone() for k, v in build_model(spec).state_dict().items()}
    else:
        spec, batch, state0, g = load_golden(tag)
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    step = FusedStep(model, arena)
    _fused(model, arena, step, batch)                      # one training step: the running statistics move
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    with pytest.raises(RuntimeError, match='EVALUATION'):
        step.infer(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi)
    model.eval()
    model.ecc.set_info([gi], 1)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=0))
    with torch.no_grad():
        ref = model.ecc(embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])).clone()
    out = step.infer(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi)
    out2 = step.infer(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi)
    torch.cuda.synchronize()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    assert torch.equal(out, ref) and torch.equal(out, out2)


def test_fused_step_refuses_what_it_does_not_serve(hip):
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep, supports
    from superpoint_graph_amd.learning import graphnet
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True)
    step = FusedStep(model, arena)
    model.eval()
    from superpoint_graph_amd.learning import ecc
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    with pytest.raises(RuntimeError, match='TRAINING step'):
        step(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi, batch['label_mode'].to(DEV))
    other = torch.nn.Module()
    other.ptn = model.ptn
    other.ecc = graphnet.GraphNetwork('gru_2,f_16,r,f_13', 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).to(DEV)
    assert not supports(other)
    with pytest.raises(NotImplementedError):
        FusedStep(other, arena)


def test_fused_step_edge_less_batch_and_refusals(hip):
    """A training batch WITHOUT superedges (sub-sampled graphs may lose all edges, learning/spg.py:150-167): no filter network
    launches, zero aggregates, the filter network's gradients are exact zeros -- bit-identical to the module path; a batch with
    a single embeddable superpoint is refused like torch's BatchNorm refuses one row per channel."""
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    from superpoint_graph_amd.learning import ecc
    spec = O.ModelSpec()
    col = synth.collate_numpy([synth.scene(3, n_sp=40, n_edges=0)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef).reshape(0, 13), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    assert batch['idxn'].numel() == 0
    torch.manual_seed(1)
    state0 = {k: v.clone() for k, v in build_model(spec).state_dict().items()}
    _compare(spec, state0, batch, None, 'mean', nsteps=2)
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True)
    step = FusedStep(model, arena)
    flag = batch['clouds_flag'].clone()
    flag[1:] = -1                                    # one embeddable superpoint left
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    with pytest.raises(ValueError, match='more than 1 value per channel'):
        step(flag, batch['clouds'][:1], batch['clouds_global'][:1], gi, batch['label_mode'].to(DEV))
