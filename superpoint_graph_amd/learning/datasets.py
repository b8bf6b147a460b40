"""Dataset providers of the training CLI: `get_info(args)` / `get_datasets(args, test_seed_offset)` per dataset name,
with the reference's contracts (learning/s3dis_dataset.py:22-90, sema3d_dataset.py:20-84, vkitti_dataset.py,
custom_dataset.py): `get_info` -> {node_feats, edge_feats, class_weights, classes, inv_class_map}; `get_datasets` ->
(train, test, valid, scaler) where every dataset yields `spg.loader(entry, ...)` samples.

The HDF5 readers need `h5py` (file I/O is outside the accelerated path); `register()` adds providers -- this is how the
reference's "custom_dataset -- to write!" hook and the in-memory datasets of the tests and benchmarks plug in."""
import functools
import os

import numpy as np
import torch

from . import spg

_PROVIDERS = {}


def register(name, get_info, get_datasets):
    _PROVIDERS[name] = (get_info, get_datasets)


def provider(name):
    if name not in _PROVIDERS:
        raise NotImplementedError('Unknown dataset ' + name)
    return _PROVIDERS[name]


class ListDataset(torch.utils.data.Dataset):
    """torchnet's ListDataset: element i = load(elem_list[i])."""

    def __init__(self, elem_list, load):
        self.list, self.load = list(elem_list), load

    def __len__(self):
        return len(self.list)

    def __getitem__(self, idx):
        return self.load(self.list[idx])


def edge_feature_width(args):
    """Number of superedge features selected by --edge_attribs (s3dis_dataset.py:63-69)."""
    return sum(3 if a.split('/')[0] in ('delta_avg', 'delta_std', 'xyz') else 1 for a in args.edge_attribs.split(','))


def _weights(args, counts_fn, n_classes):
    if args.loss_weights == 'none':
        w = np.ones((n_classes,), dtype='f4')
    else:
        c = counts_fn().astype('f4')
        w = c.mean() / c
    if args.loss_weights == 'sqrt':
        w = np.sqrt(w)
    w = torch.from_numpy(w)
    return w.cuda() if args.cuda else w


def _class_count(path):
    import h5py
    with h5py.File(path, 'r') as f:
        return f['class_count'][:]


def _build(args, root, trainlist, testlist, validlist, test_seed_offset):
    """Edge-feature standardisation + the three ListDatasets (the tail of every get_datasets of the reference)."""
    scaler = None
    if args.spg_attribs01:
        trainlist, testlist, validlist, scaler = spg.scaler01(trainlist, testlist, validlist=validlist)
    extra = _loader_backend(args, root)
    mk = lambda lst, train: ListDataset(  # noqa: E731
        [spg.spg_to_graph(*t) for t in lst],
        functools.partial(spg.loader, train=train, args=args, db_path=root, **({} if train else {'test_seed_offset': test_seed_offset}), **extra))
    return mk(trainlist, True), mk(testlist, False), mk(validlist, False), scaler


def _loader_backend(args, root, store=None):
    """Where the loader takes the superpoint clouds from: on the GPU the parsed points of every scene stay resident in HBM
    and one kernel builds all clouds of a graph (`--loader_device 1`, the default with --cuda 1); otherwise the
    reference's per-superpoint host path."""
    if store is None:
        # ONE store (and therefore one HBM-resident point cache, one set of open HDF5 handles) per dataset root for the whole
        # process: get_datasets is called once per test-time sample by eval_final (learning/main.py:272), and a fresh store
        # per call would re-read and re-upload every scene each time
        store = _H5_STORES.get(root)
        if store is None:
            store = _H5_STORES[root] = spg.H5PointStore(root)
    if getattr(args, 'cuda', 0) and getattr(args, 'loader_device', 1):
        cache = getattr(args, '_device_point_cache', None)
        if cache is None or cache._store is not store:
            cache = spg.DevicePointCache(store, torch.device('cuda', torch.cuda.current_device()))
            args._device_point_cache = cache
        return {'device_cache': cache}
    return {'store': store}


_H5_STORES = {}


# ---- S3DIS (learning/s3dis_dataset.py) ---------------------------------------------------------------------------
_S3DIS_VALID = ['hallway_1.h5', 'hallway_6.h5', 'hallway_11.h5', 'office_1.h5', 'office_6.h5', 'office_11.h5', 'office_16.h5',
                'office_21.h5', 'office_26.h5', 'office_31.h5', 'office_36.h5', 'WC_2.h5', 'storage_1.h5', 'storage_5.h5',
                'conferenceRoom_2.h5', 'auditorium_1.h5']


def _s3dis_info(args):
    def counts():
        c = _class_count(args.S3DIS_PATH + '/parsed/class_count.h5')
        return c[:, [i for i in range(6) if i != args.cvfold - 1]].sum(1)
    return {'node_feats': 14 if args.pc_attribs == '' else len(args.pc_attribs), 'edge_feats': edge_feature_width(args),
            'class_weights': _weights(args, counts, 13), 'classes': 13,
            'inv_class_map': dict(enumerate(['ceiling', 'floor', 'wall', 'column', 'beam', 'window', 'door', 'table', 'chair',
                                             'bookcase', 'sofa', 'board', 'clutter']))}


def _s3dis_datasets(args, test_seed_offset=0):
    train, test, valid = [], [], []
    for area in range(1, 7):
        path = '{}/superpoint_graphs/Area_{:d}/'.format(args.S3DIS_PATH, area)
        for fname in sorted(os.listdir(path)):
            if not fname.endswith('.h5'):
                continue
            item = spg.spg_reader(args, path + fname, True)
            if area == args.cvfold:
                test.append(item)
            elif args.use_val_set and fname in _S3DIS_VALID:
                valid.append(item)
            else:
                train.append(item)
    return _build(args, args.S3DIS_PATH, train, test, valid, test_seed_offset)


# ---- Semantic3D (learning/sema3d_dataset.py) ----------------------------------------------------------------------
_SEMA3D_TRAIN = ['bildstein_station1', 'bildstein_station5', 'domfountain_station1', 'domfountain_station3', 'neugasse_station1',
                 'sg27_station1', 'sg27_station2', 'sg27_station5', 'sg27_station9', 'sg28_station4', 'untermaederbrunnen_station1']
_SEMA3D_VALID = ['bildstein_station3', 'domfountain_station2', 'sg27_station4', 'untermaederbrunnen_station3']


def _sema3d_info(args):
    return {'node_feats': 14 if args.pc_attribs == '' else len(args.pc_attribs), 'edge_feats': edge_feature_width(args),
            'class_weights': _weights(args, lambda: _class_count(args.SEMA3D_PATH + '/parsed/class_count.h5'), 8), 'classes': 8,
            'inv_class_map': dict(enumerate(['terrain_man', 'terrain_nature', 'veget_hi', 'veget_low', 'building', 'scape',
                                             'artefact', 'cars']))}


def _sema3d_datasets(args, test_seed_offset=0):
    root = args.SEMA3D_PATH
    names = _SEMA3D_TRAIN + (_SEMA3D_VALID if args.db_train_name == 'trainval' else [])
    sets = {'train': ['train/' + n for n in names], 'valid': ['train/' + n for n in _SEMA3D_VALID] if args.use_val_set else [], 'test': []}
    sub = {'testred': 'test_reduced', 'testfull': 'test_full'}.get(args.db_test_name)
    if sub is not None:
        sets['test'] = [sub + '/' + os.path.splitext(f)[0] for f in os.listdir(root + '/superpoint_graphs/' + sub)]
    read = lambda lst: [spg.spg_reader(args, root + '/superpoint_graphs/' + n + '.h5', True) for n in lst]  # noqa: E731
    return _build(args, root, read(sets['train']), read(sets['test']), read(sets['valid']), test_seed_offset)


# ---- vKITTI (learning/vkitti_dataset.py) --------------------------------------------------------------------------
def _vkitti_info(args):
    def counts():
        c = _class_count(args.VKITTI_PATH + '/parsed/class_count.h5')
        return c[:, [i for i in range(6) if i != args.cvfold - 1]].sum(1)
    return {'node_feats': 9 if args.pc_attribs == '' else len(args.pc_attribs), 'edge_feats': edge_feature_width(args),
            'classes': 13, 'class_weights': _weights(args, counts, 13),
            'inv_class_map': dict(enumerate(['Terrain', 'Tree', 'Vegetation', 'Building', 'Road', 'GuardRail', 'TrafficSign',
                                             'TrafficLight', 'Pole', 'Misc', 'Truck', 'Car', 'Van']))}


def _vkitti_datasets(args, test_seed_offset=0):
    train, test, valid = [], [], []
    for n in range(1, 7):
        path = '{}/superpoint_graphs/0{:d}/'.format(args.VKITTI_PATH, n)
        for fname in sorted(os.listdir(path)):
            if fname.endswith('.h5'):
                (test if n == args.cvfold else train).append(spg.spg_reader(args, path + fname, True))
    return _build(args, args.VKITTI_PATH, train, test, valid, test_seed_offset)


# ---- custom_dataset template (learning/custom_dataset.py): train/ and test/ folders ------------------------------
def _custom_info(args):
    n = int(getattr(args, 'custom_classes', 10))
    return {'node_feats': 11 if args.pc_attribs == '' else len(args.pc_attribs), 'edge_feats': edge_feature_width(args),
            'class_weights': _weights(args, lambda: np.ones(n), n), 'classes': n, 'inv_class_map': {i: 'class_%d' % i for i in range(n)}}


def _custom_datasets(args, test_seed_offset=0):
    root = args.CUSTOM_SET_PATH
    read = lambda sub: [spg.spg_reader(args, root + '/superpoint_graphs/' + sub + '/' + f, True)  # noqa: E731
                        for f in sorted(os.listdir(root + '/superpoint_graphs/' + sub)) if f.endswith('.h5')]
    return _build(args, root, read('train'), read('test'), [], test_seed_offset)


register('s3dis', _s3dis_info, _s3dis_datasets)
register('sema3d', _sema3d_info, _sema3d_datasets)
register('vkitti', _vkitti_info, _vkitti_datasets)
register('custom_dataset', _custom_info, _custom_datasets)


# ---- in-memory datasets (tests, benchmarks, BASELINE config 1 without files) ---------------------------------------
def register_memory_dataset(name, train, test, valid, points, n_classes, node_feats, class_names=None):
    """`train` / `test` / `valid`: lists of (scene name, {graph-file dataset name: array}) -- the content of
    superpoint_graphs/<scene>.h5; `points`: {scene name: {superpoint id: float array [n, ncols]}} = the content of
    parsed/<scene>.h5."""
    store = spg.MemoryPointStore(points)

    def info(args):
        return {'node_feats': node_feats if args.pc_attribs == '' else len(args.pc_attribs), 'edge_feats': edge_feature_width(args),
                'class_weights': _weights(args, lambda: np.ones(n_classes), n_classes), 'classes': n_classes,
                'inv_class_map': {i: (class_names[i] if class_names else 'class_%d' % i) for i in range(n_classes)}}

    def datasets(args, test_seed_offset=0):
        read = lambda lst: [spg.spg_from_arrays(args, g, n) for n, g in lst]  # noqa: E731
        tr, te, va = read(train), read(test), read(valid)
        scaler = None
        if args.spg_attribs01:
            tr, te, va, scaler = spg.scaler01(tr, te, validlist=va)
        extra = _loader_backend(args, None, store)
        mk = lambda lst, is_train: ListDataset(  # noqa: E731
            [spg.spg_to_graph(*t) for t in lst],
            functools.partial(spg.loader, train=is_train, args=args, db_path=None,
                              **({} if is_train else {'test_seed_offset': test_seed_offset}), **extra))
        return mk(tr, True), mk(te, False), mk(va, False), scaler

    register(name, info, datasets)
