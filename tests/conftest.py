import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(tag):
    """-> (spec, batch dict of torch tensors, state0 dict, npz)"""
    from oracle import spg_oracle as O
    g = np.load(os.path.join(GOLDEN, tag + '.npz'))
    ints = g['spec/ints']
    spec = O.ModelSpec(model_config=str(g['spec/model_config']), node_feats=int(ints[0]), edge_feats=int(ints[1]),
                       ptn_widths=(tuple(int(v) for v in g['spec/ptn_widths0']), tuple(int(v) for v in g['spec/ptn_widths1'])),
                       ptn_widths_stn=(tuple(int(v) for v in g['spec/ptn_widths_stn0']), tuple(int(v) for v in g['spec/ptn_widths_stn1'])),
                       ptn_nfeat_stn=int(ints[2]), fnet_widths=tuple(int(v) for v in g['spec/fnet_widths']),
                       fnet_llbias=int(ints[3]), fnet_orthoinit=int(ints[4]), fnet_bnidx=int(ints[5]), ptn_npts=int(ints[6]))
    batch = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('batch/')}
    state0 = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('state0/')}
    return spec, batch, state0, g


def build_model(spec, state=None, n_classes=None):
    """The product model (superpoint_graph_amd.learning) for a ModelSpec, as learning/main.py:create_model does."""
    from superpoint_graph_amd.learning import graphnet, pointnet
    model = torch.nn.Module()
    nfeat = spec.ptn_widths[1][-1]
    model.ecc = graphnet.GraphNetwork(spec.model_config, nfeat, [spec.edge_feats] + list(spec.fnet_widths),
                                      spec.fnet_orthoinit, spec.fnet_llbias, spec.fnet_bnidx, 30000, use_pyg=0, cuda=1)
    model.ptn = pointnet.PointNet(list(spec.ptn_widths[0]), list(spec.ptn_widths[1]), list(spec.ptn_widths_stn[0]),
                                  list(spec.ptn_widths_stn[1]), spec.node_feats, spec.ptn_nfeat_stn,
                                  prelast_do=spec.ptn_prelast_do)
    if state is not None:
        model.load_state_dict(state)
    return model


def noise_grad(key, grads_ref):
    """True for gradient tensors that carry no signal.  A Linear / Conv1d bias directly in front of a train-mode BatchNorm
    has an analytically ZERO gradient (the normalisation removes any constant), so both sides hold round-off -- the
    reference's arithmetic leaves up to ~1e-5 of the neighbouring gradients there, the HIP path writes exact zeros.
    '<seq>.<i>.bias' is such a bias when '<seq>.<i+1>.weight' exists and is one-dimensional (a BatchNorm weight).
    Otherwise: anything whose reference maximum is below 1e-6 -- absolutely, or relative to the largest gradient of the model (a
    sum-reduced loss scales every gradient by the number of labelled superpoints: the BatchNorm shift in front of the STN's
    max-pool then carries 1e-6 of cancellation residue next to siblings of size 1 - 25)."""
    import re
    m = re.match(r'^(.*\.)(\d+)\.bias$', key)
    if m:
        nxt = grads_ref.get(f'{m.group(1)}{int(m.group(2)) + 1}.weight')
        own = grads_ref.get(f'{m.group(1)}{m.group(2)}.weight')
        if nxt is not None and nxt.dim() == 1 and own is not None and own.dim() > 1:
            return True
    top = max((float(v.abs().max()) for v in grads_ref.values() if v is not None and v.numel()), default=0.0)
    return float(grads_ref[key].abs().max()) < max(1e-6, 1e-6 * top)


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_elementwise(a, ref, rtol=1e-4, atol_frac=1e-5, what=''):
    """Element-wise form of the north-star bound: |a - ref| <= rtol * |ref| + atol for EVERY element, with the absolute
    floor atol = atol_frac * max|ref| for elements near zero (a relative bound alone is undefined there)."""
    a, ref = a.detach().double().cpu(), (ref if torch.is_tensor(ref) else torch.from_numpy(np.asarray(ref))).detach().double().cpu()
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    bound = rtol * ref.abs() + atol_frac * float(ref.abs().max())
    excess = ((a - ref).abs() - bound).max()
    assert float(excess) <= 0.0, f'{what}: worst element exceeds rtol {rtol} + {atol_frac} * max|ref| by {float(excess):.3e}'


@pytest.fixture(scope='session')
def hip():
    """The loaded HIP library; building it first if the .so is absent (build container)."""
    from superpoint_graph_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()
