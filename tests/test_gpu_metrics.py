"""GPU: evaluation accounting on the device (spg_eval_accumulate) against the reference's confusion matrix / scores
(tests/golden/metrics.npz, produced by the imported reference) -- integers and derived float64 scores bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import spg_metrics_oracle as MO

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('tag', ['multi', 'single'])
def test_device_confusion_matrix_bit_exact(hip, tag):
    from superpoint_graph_amd.learning import metrics
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    smp = g['samples'] if tag == 'multi' else g['samples'][:1]
    logits = torch.from_numpy(smp).to(DEV) if tag == 'multi' else torch.from_numpy(smp[0]).to(DEV)
    lv, lm = torch.from_numpy(g['label_vec']).to(DEV), torch.from_numpy(g['label_mode']).to(DEV)
    m = metrics.ConfusionMatrix(13)
    pred = m.count_predicted_batch_device(lv, logits, lm)
    assert np.array_equal(pred.cpu().numpy(), g[f'{tag}/pred'])
    assert np.array_equal(m.confusion_matrix, g[f'{tag}/cm'])
    assert m.accuracy_counts() == (int(g[f'{tag}/correct']), int(g[f'{tag}/counted']))
    assert np.array_equal(np.array(m.get_intersection_union_per_class()), g[f'{tag}/iou'])
    assert m.get_overall_accuracy() == float(g[f'{tag}/oa']) and m.get_average_intersection_union() == float(g[f'{tag}/miou'])
    assert m.get_mean_class_accuracy() == float(g[f'{tag}/mca'])
    # streaming: a second batch accumulates
    m.count_predicted_batch_device(lv, logits, lm)
    assert np.array_equal(m.confusion_matrix, 2 * g[f'{tag}/cm'])


def test_large_random_vs_oracle(hip):
    from superpoint_graph_amd.learning import metrics
    rng = np.random.default_rng(9)
    N, C, S = 50000, 8, 10
    samples = [rng.normal(size=(N, C)).astype(np.float32) for _ in range(S)]
    lv = (rng.integers(0, 10000, size=(N, C)) * (rng.random((N, C)) < 0.2)).astype(np.int64)
    lm = np.where(lv.sum(1) == 0, -100, lv.argmax(1)).astype(np.int64)
    pred, cm, correct, counted = MO.aggregate(samples, lm, lv, C)
    m = metrics.ConfusionMatrix(C)
    p = m.count_predicted_batch_device(torch.from_numpy(lv).to(DEV), torch.from_numpy(np.stack(samples)).to(DEV),
                                       torch.from_numpy(lm).to(DEV))
    assert np.array_equal(p.cpu().numpy(), pred) and np.array_equal(m.confusion_matrix, cm)
    assert m.accuracy_counts() == (correct, counted)
