"""CPU, build container only (skipped where /root/reference is absent): the reference's own learning/main.py builds
its model out of THIS package's modules after the switch documented in INTEGRATION.md (sys.modules aliasing), and the
result is checkpoint-compatible with the reference (state_dict keys, shapes, seed-for-seed initial values)."""
import os
import sys
import types

import pytest
import torch

from conftest import load_golden

REF = os.environ.get('SPG_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'learning')), reason='reference checkout not present')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


@pytest.fixture()
def ref_main():
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    try:
        # loader-side dependencies that are not installed offline and are irrelevant to model creation
        for name in ('h5py', 'igraph', 'torchnet', 'transforms3d', 'transforms3d.zooms', 'transforms3d.axangles'):
            if name not in sys.modules:
                _stub(name)
        import superpoint_graph_amd.learning as amd
        from superpoint_graph_amd.learning import ecc, graphnet, modules, pointnet
        sys.path.insert(0, REF)
        for k in [k for k in sys.modules if k == 'learning' or k.startswith('learning.')]:
            del sys.modules[k]
        import learning                                   # the reference package (its __init__ edits sys.path only)
        # the two-line switch of INTEGRATION.md
        sys.modules['learning.pointnet'] = pointnet; sys.modules['learning.graphnet'] = graphnet
        sys.modules['learning.modules'] = modules; sys.modules['learning.ecc'] = ecc
        learning.pointnet, learning.graphnet, learning.modules, learning.ecc = pointnet, graphnet, modules, ecc
        sys.modules['ecc'] = ecc                          # learning/spg.py does `import ecc` after learning/__init__'s path edit
        from learning import main as ref_main_mod
        yield ref_main_mod
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.modules.update(saved)


def _s3dis_args():
    import argparse
    return argparse.Namespace(model_config='gru_10_0,f_13', ptn_widths=[[64, 64, 128, 128, 256], [256, 64, 32]],
                                 ptn_widths_stn=[[64, 64, 128], [128, 64]], ptn_nfeat_stn=14, ptn_prelast_do=0,
                                 fnet_widths=[32, 128, 64], fnet_orthoinit=1, fnet_llbias=0, fnet_bnidx=2,
                                 edge_mem_limit=30000, use_pyg=0, cuda=0, optim='adam', lr=1e-2, wd=0, momentum=0.9)


def test_reference_create_model_builds_amd_modules(ref_main):
    from superpoint_graph_amd.learning import graphnet, pointnet
    args = _s3dis_args()
    dbinfo = dict(node_feats=14, edge_feats=13, classes=13)
    ref_main.set_seed(1, cuda=False)
    model = ref_main.create_model(args, dbinfo)           # reference code, learning/main.py:414-431
    assert isinstance(model.ecc, graphnet.GraphNetwork) and isinstance(model.ptn, pointnet.PointNet)
    assert sum(p.numel() for p in model.parameters()) == 279409
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(state0.keys())
    for k in ('ecc.0._cell.weight_ih', 'ecc.0._fnet.0.weight', 'ecc.0._fnet.7.weight', 'ecc.1.weight', 'ptn.convs.0.weight',
              'ptn.stn.convs.6.weight', 'ptn.fcs.6.weight'):
        assert torch.equal(sd[k], state0[k]), k           # seed-for-seed identical to the reference's own init
    opt = ref_main.create_optimizer(args, model)          # learning/main.py:433-437
    assert len(opt.param_groups[0]['params']) == len(list(model.parameters()))
    # resume(): old checkpoints carry 4 InstanceNorm running-stat keys that main.py:403 filters out
    ckpt = dict(state0)
    for k in ('ecc.0._cell.inh.running_mean', 'ecc.0._cell.inh.running_var', 'ecc.0._cell.ini.running_mean', 'ecc.0._cell.ini.running_var'):
        ckpt[k] = torch.zeros(1)
    model.load_state_dict({k: ckpt[k] for k in ckpt if k not in ['ecc.0._cell.inh.running_mean', 'ecc.0._cell.inh.running_var',
                                                                 'ecc.0._cell.ini.running_mean', 'ecc.0._cell.ini.running_var']})


def test_reference_filter_valid_and_collate_contract(ref_main):
    out = torch.randn(6, 13)
    tgt = torch.tensor([1, -100, 3, 4, -100, 0])
    o, t = ref_main.filter_valid(out, tgt)                # learning/main.py:447-452
    assert o.shape == (4, 13) and t.tolist() == [1, 3, 4, 0]


def test_loader_and_dataset_plumbing_bit_identical_to_reference():
    """spg_reader body / spg_edge_features / scaler01 / spg_to_graph / loader (neighbourhood sub-sampling, hard cut-off,
    load_superpoint, augment_cloud with scale + rotation + mirroring + jitter) of this package against the reference's own
    learning/spg.py on the in-memory dataset of tests/main_fixture.py, same seeds: every array of every sample identical."""
    import random
    import subprocess
    import sys as _sys
    code = r'''
import sys, random, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import main_fixture
import gen_main_golden as G
G.install_shims()
sys.path.insert(0, G.REF)
import learning
train, test = main_fixture.make_dataset(0)
G.install_dataset(train, test)
from superpoint_graph_amd.learning import main as cli, datasets
args = cli.parse_args(['--dataset', 'x'] + main_fixture.CLI); args.cuda = 0
import custom_dataset
rtr, rte, _, rsc = custom_dataset.get_datasets(args)
points = {n: p for n, _, p in train + test}
datasets.register_memory_dataset('memtest', [(n, g) for n, g, _ in train], [(n, g) for n, g, _ in test], [], points, 13, 14)
atr, ate, _, asc = datasets.provider('memtest')[1](args)
assert np.array_equal(rsc.mean_, asc.mean_) and np.array_equal(rsc.scale_, asc.scale_)
n = 0
for ds_r, ds_a in ((rtr, atr), (rte, ate)):
    for i in range(len(ds_r)):
        random.seed(5); np.random.seed(5); r = ds_r[i]
        random.seed(5); np.random.seed(5); a = ds_a[i]
        for x, y in ((r[0], a[0]), (r[3], a[3]), (r[4], a[4]), (r[5], a[5])):
            assert np.array_equal(np.asarray(x), np.asarray(y))
        assert r[2] == a[2] and r[1].get_edgelist() == a[1].get_edgelist()
        assert np.array_equal(np.asarray(r[1].es.get_attribute_values('f')), np.asarray(a[1].es.get_attribute_values('f')))
        n += 1
print('identical samples:', n)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    out = subprocess.run([_sys.executable, '-c', code], capture_output=True, text=True, timeout=600)   # own process: sys.modules shims stay out of this one
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'identical samples: 6' in out.stdout
