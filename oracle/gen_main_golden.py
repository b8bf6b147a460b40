#!/usr/bin/env python3
"""ORACLE / test infrastructure (never the product path): runs the REFERENCE's own `learning/main.py` -- its argparse,
dataset plumbing (`spg_reader`, `spg_edge_features`, `scaler01`, `spg_to_igraph`, `loader` with neighbourhood
sub-sampling, `load_superpoint`, `augment_cloud`), `create_model`, the train / eval / eval_final loops, Adam, the
checkpoint writer -- on the CPU (`--cuda 0`, BASELINE.json configs[0] plumbing) over the small in-memory dataset of
tests/main_fixture.py, and writes what it printed / saved to tests/golden/main_cli.npz.

Needs /root/reference (build container only).  Packages that are not installable offline are replaced by shims
(SURVEY.md App. C): `h5py` -> in-memory files, `torchnet` -> ListDataset + the two meters (same arithmetic; the loss
meter also records every value), `igraph` -> superpoint_graph_amd.learning.spg.SuperpointGraph (the igraph-API subset the
loader touches), `transforms3d` -> its closed forms (oracle/spg_loader_oracle.py).  The reference's matrix-filter
ECC backward raises on torch >= 1.5 (GraphConvModule.py:146): `GraphConvFunction` is replaced by the restated
oracle.EccFunction exactly as in oracle/validate_against_reference.py, where that restatement is pinned.

    python oracle/gen_main_golden.py                      # writes tests/golden/main_cli.npz
    python oracle/gen_main_golden.py --variant frozen     # lr 1e-7: tests/golden/main_cli_frozen.npz
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('SPG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import main_fixture  # noqa: E402
from oracle import spg_loader_oracle as L  # noqa: E402
from oracle import spg_oracle as O  # noqa: E402

DB = '/memdb'
FILES = {}            # path -> {dataset name: array}
WRITTEN = {}          # path -> {dataset name: array}   (predictions_*.h5)
LOSSES = []           # every value added to an AverageValueMeter, in order


class _DS:
    def __init__(self, a):
        self.a = np.asarray(a)
        self.shape, self.size = self.a.shape, self.a.size

    def __getitem__(self, k):
        return self.a[k]


class _H5File:
    def __init__(self, path, mode='r'):
        self.path, self.mode = os.path.normpath(path), mode
        if mode == 'w':
            WRITTEN[self.path] = {}
        elif self.path not in FILES:
            raise FileNotFoundError(path)

    def __getitem__(self, k):
        return _DS(FILES[self.path][k])

    def keys(self):
        return FILES[self.path].keys()

    def create_dataset(self, name, data):
        WRITTEN[self.path][name] = np.asarray(data)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _ListDataset(torch.utils.data.Dataset):
    def __init__(self, elem_list, load):
        self.list, self.load = elem_list, load

    def __len__(self):
        return len(self.list)

    def __getitem__(self, i):
        return self.load(self.list[i])


def install_shims():
    from superpoint_graph_amd.learning import meters, spg as amd_spg
    h5 = types.ModuleType('h5py')
    h5.File = _H5File
    sys.modules['h5py'] = h5

    class Avg(meters.AverageValueMeter):
        def add(self, value, n=1):
            LOSSES.append(float(value))
            super().add(value, n)
    tnt = types.ModuleType('torchnet')
    tnt.dataset = types.SimpleNamespace(ListDataset=_ListDataset)
    tnt.meter = types.SimpleNamespace(AverageValueMeter=Avg, ClassErrorMeter=meters.ClassErrorMeter)
    sys.modules['torchnet'] = tnt
    ig = types.ModuleType('igraph')
    ig.Graph = amd_spg.SuperpointGraph
    sys.modules['igraph'] = ig
    t3 = types.ModuleType('transforms3d')
    t3.zooms = types.SimpleNamespace(zfdir2mat=lambda s, d=None: L.zoom(s) if d is None else (L.mirror(0) if list(d) == [1, 0, 0] else L.mirror(1)))
    t3.axangles = types.SimpleNamespace(axangle2mat=lambda ax, a: L.rot_z(a))
    sys.modules['transforms3d'] = t3
    sys.modules['transforms3d.zooms'], sys.modules['transforms3d.axangles'] = t3.zooms, t3.axangles


def install_dataset(train, test):
    """The reference's dataset-module contract (custom_dataset.py:27-74) over the in-memory files, built from the
    REFERENCE's own spg functions."""
    import functools
    for name, graph, points in train + test:
        FILES[os.path.normpath('{}/superpoint_graphs/{}.h5'.format(DB, name))] = graph
        FILES[os.path.normpath('{}/parsed/{}.h5'.format(DB, name))] = {'{:d}'.format(k): v for k, v in points.items()}
    import torchnet as tnt
    from learning import spg

    def get_datasets(args, test_seed_offset=0):
        trainlist = [spg.spg_reader(args, '{}/superpoint_graphs/{}.h5'.format(DB, n), True) for n, _, _ in train]
        testlist = [spg.spg_reader(args, '{}/superpoint_graphs/{}.h5'.format(DB, n), True) for n, _, _ in test]
        trainlist, testlist, validlist, scaler = spg.scaler01(trainlist, testlist)
        return (tnt.dataset.ListDataset([spg.spg_to_igraph(*t) for t in trainlist],
                                        functools.partial(spg.loader, train=True, args=args, db_path=DB)),
                tnt.dataset.ListDataset([spg.spg_to_igraph(*t) for t in testlist],
                                        functools.partial(spg.loader, train=False, args=args, db_path=DB, test_seed_offset=test_seed_offset)),
                tnt.dataset.ListDataset([], None), scaler)

    def get_info(args):
        edge_feats = sum(3 if a.split('/')[0] in ('delta_avg', 'delta_std', 'xyz') else 1 for a in args.edge_attribs.split(','))
        return {'node_feats': len(args.pc_attribs), 'edge_feats': edge_feats, 'class_weights': torch.ones(main_fixture.N_CLASSES),
                'classes': main_fixture.N_CLASSES, 'inv_class_map': {i: 'class_%d' % i for i in range(main_fixture.N_CLASSES)}}
    mod = types.ModuleType('custom_dataset')
    mod.get_datasets, mod.get_info = get_datasets, get_info
    sys.modules['custom_dataset'] = mod


def main():
    assert os.path.isdir(os.path.join(REF, 'learning')), 'reference checkout not found'
    install_shims()
    sys.path.insert(0, REF)
    import learning  # noqa: F401  (its __init__ puts learning/ on sys.path)
    from learning import ecc
    ecc.GraphConvFunction = O.EccFunction                    # see the module docstring
    import learning.ecc.GraphConvModule as gcm
    gcm.GraphConvFunction = O.EccFunction
    import learning.modules as refmodules
    refmodules.ecc.GraphConvFunction = O.EccFunction
    train, test = main_fixture.make_dataset(0)
    install_dataset(train, test)
    from learning import main as ref_main
    odir = tempfile.mkdtemp(prefix='spg_main_golden_')
    torch.set_num_threads(8)
    # `--variant frozen`: the same run with lr = 1e-7 -- the parameters move by <= 4e-7 in four Adam steps, so EVERY loss of the run
    # (not only the first) is comparable at fp32 round-off and the end-to-end test can be sharp (tests/golden/main_cli_frozen.npz)
    frozen = len(sys.argv) > 2 and sys.argv[1] == '--variant' and sys.argv[2] == 'frozen'
    cli = list(main_fixture.CLI)
    if frozen:
        cli[cli.index('--lr') + 1] = '1e-7'
    argv = ['main.py', '--dataset', 'custom_dataset', '--cuda', '0', '--odir', odir] + cli
    old = sys.argv
    sys.argv = argv
    try:
        ref_main.main()
    finally:
        sys.argv = old
    with open(os.path.join(odir, 'trainlog.json')) as f:
        stats = json.load(f)
    with open(os.path.join(odir, 'scores_test.json')) as f:
        scores = json.load(f)[0]
    cm = np.load(os.path.join(odir, 'pointwise_cm.npy'))
    ckpt = torch.load(os.path.join(odir, 'model.pth.tar'), weights_only=False)
    preds = WRITTEN[os.path.normpath(os.path.join(odir, 'predictions_test.h5'))]
    out = {'losses': np.array(LOSSES), 'stats_json': np.array(json.dumps(stats)), 'scores_json': np.array(json.dumps(scores)),
           'pointwise_cm': cm, 'scaler_mean': ckpt['scaler'].mean_, 'scaler_scale': ckpt['scaler'].scale_}
    for k, v in preds.items():
        out['pred/' + k] = v
    for k in ('ecc.1.weight', 'ecc.0._cell.weight_ih', 'ecc.0._fnet.7.weight', 'ptn.convs.12.weight', 'ptn.fcs.6.weight', 'ptn.stn.proj.weight',
              'ptn.convs.13.running_mean', 'ptn.convs.13.running_var', 'ecc.0._fnet.5.running_var'):
        out['param/' + k] = ckpt['state_dict'][k].numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'main_cli_frozen.npz' if frozen else 'main_cli.npz')
    np.savez_compressed(path, **out)
    print('losses:', LOSSES)
    print('stats:', json.dumps(stats))
    print('scores:', json.dumps(scores))
    print('wrote', path)


if __name__ == '__main__':
    main()
