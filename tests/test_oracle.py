"""CPU: the oracle (oracle/spg_oracle.py) against the golden vectors generated from the imported reference
(oracle/validate_against_reference.py), plus the reference's own property tests restated
(learning/ecc/test_GraphConvModule.py:23-75)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxrel
from oracle import spg_oracle as O


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'])
def test_oracle_eval_forward_matches_reference(tag):
    spec, batch, state0, g = load_golden(tag)
    emb, logits = O.model_forward(batch, spec, {k: v.clone() for k, v in state0.items()}, False)
    assert maxrel(emb, torch.from_numpy(g['eval/emb'])) < 2e-6
    assert maxrel(logits, torch.from_numpy(g['eval/logits'])) < 5e-6


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'])
def test_oracle_train_step_matches_reference(tag):
    spec, batch, state0, g = load_golden(tag)
    st = {k: v.clone() for k, v in state0.items()}
    cw = torch.from_numpy(g['class_weights'])
    loss, logits, emb, grads = O.train_step(batch, spec, st, cw)
    assert maxrel(emb, torch.from_numpy(g['train/emb'])) < 2e-5
    assert maxrel(logits, torch.from_numpy(g['train/logits'])) < 5e-5
    assert maxrel(loss, torch.from_numpy(g['train/loss'])) < 1e-5
    for k in [k[5:] for k in g.files if k.startswith('grad/')]:
        ref = torch.from_numpy(g['grad/' + k])
        if float(ref.abs().max()) < 1e-6:      # bias in front of a train-mode BatchNorm: rounding noise on both sides
            continue
        assert maxrel(grads[k], ref) < 2e-4, k
    for k in [k[7:] for k in g.files if k.startswith('state1/')]:
        assert maxrel(st[k].double(), torch.from_numpy(g['state1/' + k]).double()) < 1e-5, k


def test_index_contract_bit_exact():
    for tag in ('s3dis_gru10_matrix', 'vector_gru4_small', 'lstm3_matrix_small'):
        spec, batch, state0, g = load_golden(tag)
        n_graphs = len([k for k in g.files if k.startswith('graph/') and k.endswith('/n')])
        el = [g[f'graph/{i}/edges'] for i in range(n_graphs)]
        vc = [int(g[f'graph/{i}/n']) for i in range(n_graphs)]
        ef = [g[f'graph/{i}/feats'] for i in range(n_graphs)]
        idxn, degs, edgefeats, edge_indexes = O.set_batch(el, vc, ef)
        assert np.array_equal(idxn, g['batch/idxn']) and np.array_equal(degs, g['batch/degs'])
        assert np.array_equal(edge_indexes, g['batch/edge_indexes']) and np.array_equal(edgefeats, g['batch/edgefeats'])
        rp = O.csr_by_target(degs)
        assert rp[-1] == len(idxn)
        rrp, order = O.csr_by_source(idxn, len(degs))
        assert np.array_equal(np.sort(order), np.arange(len(idxn)))
        for j in range(len(degs)):
            seg = order[rrp[j]:rrp[j + 1]]
            assert np.all(idxn[seg] == j) and np.all(np.diff(seg) > 0)


def test_ops_golden():
    g = np.load('tests/golden/ops.npz') if False else np.load(__import__('os').path.join(__import__('conftest').GOLDEN, 'ops.npz'))
    x, w = torch.from_numpy(g['ecc_x']), torch.from_numpy(g['ecc_w'])
    idxn, degs = torch.from_numpy(g['ecc_idxn']), torch.from_numpy(g['ecc_degs'])
    out = O.ecc_forward(x, w, idxn, degs)
    assert maxrel(out, torch.from_numpy(g['ecc_out'])) < 1e-14
    assert float(out[1].abs().max()) == 0.0                 # zero in-degree row is exactly zero
    gx, gw = O.ecc_backward(torch.from_numpy(g['eccv_x']), torch.from_numpy(g['eccv_w']), torch.from_numpy(g['eccv_go']), idxn, degs)
    assert maxrel(gx, torch.from_numpy(g['eccv_gx'])) < 1e-14 and maxrel(gw, torch.from_numpy(g['eccv_gw'])) < 1e-14
    P = {'c.' + k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('gru_p/')}
    hy = O.gru_cell_ex(torch.from_numpy(g['gru_in']), torch.from_numpy(g['gru_h']), P, 'c')
    assert maxrel(hy, torch.from_numpy(g['gru_out'])) < 2e-6


def test_gradcheck_and_shard_invariance():
    # learning/ecc/test_GraphConvModule.py:29-36 fixture
    gen = torch.Generator().manual_seed(0)
    n, e, cin, cout = 20, 50, 10, 15
    x = torch.randn(n, cin, generator=gen, dtype=torch.float64, requires_grad=True)
    w = torch.randn(e, cin, cout, generator=gen, dtype=torch.float64, requires_grad=True)
    idxn = torch.randint(0, n, (e,), generator=gen)
    degs = torch.LongTensor([5, 0, 15, 20, 10])
    assert torch.autograd.gradcheck(lambda a, b: O.EccFunction.apply(a, b, cin, cout, idxn, None, degs, None, 30), (x, w))
    for lim in (1, 30, 1e10):
        sh = O.get_edge_shards(degs.numpy(), lim)
        assert sum(a for a, _ in sh) == 5 and sum(b for _, b in sh) == 50


def test_loader_oracle_matches_reference_golden():
    """oracle/spg_loader_oracle.py against the clouds the imported reference's load_superpoint / augment_cloud produced
    (tests/golden/loader.npz, written by oracle/validate_against_reference.py::check_loader)."""
    import os
    import random as pyrandom
    from conftest import GOLDEN
    from oracle import spg_loader_oracle as L
    g = np.load(os.path.join(GOLDEN, 'loader.npz'))
    points, offsets, ids = g['points'], g['offsets'], g['ids']
    for tag, attribs, norm in (('s3dis', 'xyzrgbelpsvXYZ', 1), ('sema3d', 'xyzrgbelpsv', 1), ('nonorm', 'xyzelpsv', 0)):
        m = L.load_batch(points, offsets, ids, 40, 128, norm, attribs, train=False, test_seed_offset=3)
        for key in ('flag', 'slot', 'sample_idx', 'clouds', 'diam'):
            assert np.array_equal(m[key], g[f'{tag}/{key}']), (tag, key)
    assert list(g['s3dis/flag']) == [-1, -1, 0, 0, 0, 0, 0, -1, 0, 0]            # < ptn_minpts points: no cloud
    assert g['s3dis/diam'][2] == 0.0      # degenerate superpoint (all points equal): (x - mean) / 1e-10, rounding noise blown up
    np.random.seed(5); pyrandom.seed(6)
    m = L.load_batch(points, offsets, ids, 40, 128, 1, 'xyzrgbelpsvXYZ', train=True,
                     augm=dict(scale=1.1, rot=1, mirror_prob=1.0, jitter=1), nprandom=np.random, pyrandom=pyrandom)
    for key in ('sample_idx', 'clouds', 'diam', 'M', 'noise'):
        assert np.array_equal(m[key], g[f'train/{key}']), key
    # the summation order the device kernel relies on: numpy's float32 mean over axis 0 adds the rows one by one
    rows = points[offsets[8]:offsets[9]][g['s3dis/sample_idx'][8]][:, :3]
    acc = np.zeros(3, np.float32)
    for r in rows:
        acc = acc + r
    assert np.array_equal(acc / np.float32(128), np.mean(rows, axis=0))


def test_metrics_oracle_and_host_mirror_match_reference_golden():
    """Evaluation accounting (reference learning/metrics.py + main.py eval loops): the oracle and the host interface of the
    product's ConfusionMatrix against the values the imported reference produced (tests/golden/metrics.npz)."""
    import os
    from conftest import GOLDEN
    from oracle import spg_metrics_oracle as MO
    from superpoint_graph_amd.learning import metrics
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    samples, lv, lm = list(g['samples']), g['label_vec'], g['label_mode']
    for tag, smp in (('multi', samples), ('single', samples[:1])):
        pred, cm, correct, counted = MO.aggregate(smp, lm, lv, 13)
        assert np.array_equal(pred, g[f'{tag}/pred']) and np.array_equal(cm, g[f'{tag}/cm'])
        assert (correct, counted) == (int(g[f'{tag}/correct']), int(g[f'{tag}/counted']))
        iou, oa, miou, mca = MO.scores(cm)
        assert np.array_equal(np.array(iou), g[f'{tag}/iou']) and oa == float(g[f'{tag}/oa'])
        assert miou == float(g[f'{tag}/miou']) and mca == float(g[f'{tag}/mca'])
        # the product class, host interface (same method names as the reference)
        m = metrics.ConfusionMatrix(13)
        idx = lm != -100
        m.count_predicted_batch(lv[idx], pred[idx])
        assert np.array_equal(m.confusion_matrix, cm)
        assert m.get_intersection_union_per_class() == iou and m.get_overall_accuracy() == oa
        assert m.get_average_intersection_union() == miou and m.get_mean_class_accuracy() == mca
    assert g['multi/pred'][7] == 0          # exact tie: the first arg-max
