"""ORACLE (test infrastructure): numpy restatement of the evaluation accounting of the reference --
learning/main.py:246-263 / eval_final :267-311 (mean over the test-time samples, argmax, filter_valid :447-452) and
learning/metrics.py:16-18,31-89 (confusion-matrix update and the derived scores).  Pinned against the imported
reference `metrics.ConfusionMatrix` + the same numpy calls by oracle/validate_against_reference.py::check_metrics;
golden vectors in tests/golden/metrics.npz."""
import numpy as np


def aggregate(samples, label_mode, label_vec, n_classes):
    """samples: list of [N, C] float32 logits (one per sampling seed).  -> (pred i64 [N], confusion f64 [C, C],
    correct, counted)"""
    o = np.mean(np.stack(samples, 0), 0) if len(samples) > 1 else samples[0]      # main.py:296-299
    pred = np.argmax(o, 1)
    idx = label_mode != -100                                                      # filter_valid, main.py:447-452
    cm = np.zeros((n_classes, n_classes))
    for i in np.nonzero(idx)[0]:                                                  # metrics.py:16-18
        cm[:, pred[i]] += label_vec[i, :]
    return pred.astype(np.int64), cm, int((pred[idx] == label_mode[idx]).sum()), int(idx.sum())


def scores(cm):
    """(per-class IoU list, overall accuracy, average IoU, mean class accuracy) -- metrics.py:31-86"""
    n = cm.shape[0]
    diag = np.array([cm[i][i] for i in range(n)])
    row_err = cm.sum(1) - diag
    col_err = cm.sum(0) - diag
    div = diag + row_err + col_err
    div[diag == 0] = 1
    iou = [float(diag[i]) / div[i] for i in range(n)]
    total = cm.sum()
    oa = float(diag.sum()) / (total if total != 0 else 1)
    seen = ((cm.sum(1) + cm.sum(0)) != 0).sum()
    miou = sum(iou) / seen
    mca = sum(cm[i][i] / max(1, np.sum(cm[i, :])) for i in range(n)) / n
    return iou, oa, miou, mca
