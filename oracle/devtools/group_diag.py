"""Which gradients differ between grouped and separate launches (spg_tune key 11)?  run on the GPU box: python oracle/devtools/group_diag.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model, load_golden  # noqa: E402
from test_gpu_grouped import _step  # noqa: E402
from superpoint_graph_amd import _lib  # noqa: E402

L = _lib.lib()
spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
cw = torch.from_numpy(g['class_weights']).cuda()
res = []
for off in (1, 0, 1, 0):
    old = L.spg_tune(11, off)
    model = build_model(spec, state0).cuda().train()
    res.append(_step(model, batch, cw, None))
    L.spg_tune(11, old)
for (i, j) in ((0, 2), (1, 3), (0, 1)):
    a, b = res[i], res[j]
    bad = [(k, float((a[3][k] - b[3][k]).abs().max() / (a[3][k].abs().max() + 1e-30))) for k in a[3] if not torch.equal(a[3][k], b[3][k])]
    print('runs', i, j, 'loss equal', torch.equal(a[0], b[0]), 'differing gradients:', len(bad))
    for k, v in bad:
        print('   ', k, f'{v:.2e}')
