"""GPU: degenerate batches through the module API against the CPU oracle -- an edge-less graph, a single superpoint, a batch
whose superpoints are all too small to embed, a batch with one valid cloud -- the inputs the reference's loader can produce
(learning/spg.py:150-167: `clouds_flag` -1 for superpoints below ptn_minpts; sub-sampled graphs may lose all edges)."""
import numpy as np
import pytest
import torch

from conftest import assert_elementwise, build_model, maxrel
from oracle import spg_oracle as O
from superpoint_graph_amd import synth
from test_gpu_model import _run

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _batch(scene):
    col = synth.collate_numpy([scene])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    return dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))


@pytest.fixture(scope='module')
def model_and_state():
    spec = O.ModelSpec()
    torch.manual_seed(1)
    model = build_model(spec)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2); m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.1)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return spec, model.to(DEV).eval(), state


def _check_eval(model_and_state, batch):
    spec, model, state = model_and_state
    with torch.no_grad():
        emb, logits, _ = _run(model, batch)
    eo, lo = O.model_forward(batch, spec, state, False)
    assert emb.shape == eo.shape and logits.shape == lo.shape
    if eo.numel() and float(eo.abs().max()) > 0:
        assert_elementwise(emb, eo, what='embeddings')
    else:
        assert float(emb.abs().sum()) == 0.0
    assert_elementwise(logits, lo, what='logits')
    return emb, logits


@pytest.mark.parametrize('n_sp', [40, 1])
def test_edge_less_graph(hip, model_and_state, n_sp):
    """no superedges at all: every aggregate is zero, the GRU still runs its iterations (zero-degree rule of
    learning/ecc/GraphConvModule.py:82-88)"""
    batch = _batch(synth.scene(3, n_sp=n_sp, n_edges=0))
    assert batch['idxn'].numel() == 0 and int(batch['degs'].sum()) == 0
    _check_eval(model_and_state, batch)


def test_all_superpoints_too_small(hip, model_and_state):
    """clouds_flag -1 everywhere: no cloud is embedded, the embeddings are exact zeros (learning/pointnet.py:150-158), the
    graph network still classifies"""
    scene = synth.scene(4, n_sp=25, n_edges=90)
    batch = _batch(scene)
    batch['clouds_flag'] = torch.full_like(batch['clouds_flag'], -1)
    batch['clouds'] = batch['clouds'][:0]
    batch['clouds_global'] = batch['clouds_global'][:0]
    spec, model, state = model_and_state
    with torch.no_grad():
        emb, logits, _ = _run(model, batch)
    assert emb.shape == (25, 32) and float(emb.abs().max()) == 0.0
    # the oracle's PointNet cannot take an empty batch either (nor is it asked to: CloudEmbedder only scatters the valid
    # rows); its graph network on all-zero embeddings is the expected output
    lo = O.graph_network_forward(torch.zeros(25, 32), batch['edgefeats'], batch['idxn'], batch['degs'], spec, state, False)
    assert_elementwise(logits, lo, what='logits')


def test_single_valid_cloud(hip, model_and_state):
    """one embeddable superpoint among too-small ones: inference works; a TRAINING step is refused like torch's BatchNorm
    refuses a single row per channel (the FC head would normalise over one sample)"""
    scene = synth.scene(5, n_sp=12, n_edges=40)
    batch = _batch(scene)
    flag = torch.full_like(batch['clouds_flag'], -1)
    keep = int((batch['clouds_flag'] == 0).nonzero()[0])
    pos = int((batch['clouds_flag'][:keep] == 0).sum())
    flag[keep] = 0
    batch['clouds_flag'] = flag
    batch['clouds'] = batch['clouds'][pos:pos + 1].contiguous()
    batch['clouds_global'] = batch['clouds_global'][pos:pos + 1].contiguous()
    emb, _ = _check_eval(model_and_state, batch)
    assert int((emb.abs().sum(1) > 0).sum()) == 1
    spec, model, state = model_and_state
    model.train()
    try:
        with pytest.raises((RuntimeError, ValueError)):
            _run(model, batch)
    finally:
        model.eval()
        model.load_state_dict(state)
