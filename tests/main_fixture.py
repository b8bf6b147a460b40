"""Small in-memory superpoint-graph dataset for the end-to-end CLI tests (train 2 epochs + eval + multi-sample final
evaluation): shared by oracle/gen_main_golden.py (runs the REFERENCE's learning/main.py on it, CPU) and
tests/test_gpu_main.py (runs superpoint_graph_amd.learning.main on it, GPU).  Everything derives from one seed.

A scene = what partition/ writes for one room: the superpoint-graph file content (labels histogram, centroids, length /
surface / volume / point count per superpoint, directed superedges with delta mean / std) and the parsed point rows
[n, 15] of every superpoint."""
import numpy as np

N_CLASSES = 13
CLI = ['--epochs', '2', '--batch_size', '2', '--nworkers', '0', '--lr', '1e-2', '--model_config', 'gru_10_0,f_13',
       '--ptn_nfeat_stn', '14', '--pc_attribs', 'xyzrgbelpsvXYZ', '--ptn_minpts', '10', '--spg_augm_nneigh', '6',
       '--spg_augm_order', '2', '--spg_augm_hardcutoff', '30', '--test_multisamp_n', '2', '--test_nth_epoch', '1',
       '--db_test_name', 'test', '--seed', '3', '--pc_augm_scale', '1.1', '--pc_augm_mirror_prob', '0.4']


def make_scene(rng, name, n_sp):
    counts = np.clip(np.round(rng.lognormal(np.log(60.0), 1.0, n_sp)), 4, 400).astype(np.int64)
    labels = rng.integers(0, N_CLASSES, n_sp)
    sp_labels = np.zeros((n_sp, 1 + N_CLASSES), dtype=np.uint32)
    for i in range(n_sp):
        if rng.random() < 0.08:
            sp_labels[i, 0] = counts[i]                           # unlabelled superpoint
        else:
            main = int(counts[i] * 0.8)
            sp_labels[i, 1 + labels[i]] = main
            sp_labels[i, 1 + int(rng.integers(0, N_CLASSES))] += counts[i] - main
    cent = rng.uniform(0.0, 6.0, (n_sp, 3)).astype(np.float32)
    d2 = ((cent[:, None] - cent[None]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nbr = np.argsort(d2, axis=1)[:, :4]
    pairs = sorted({(min(i, int(j)), max(i, int(j))) for i in range(n_sp) for j in nbr[i, :int(rng.integers(1, 4))]})
    pairs = np.array(pairs, dtype=np.uint32)
    pairs = pairs[rng.permutation(len(pairs))]
    edges = np.concatenate([pairs, pairs[:, ::-1]], 0)            # both directions, as partition/graphs.py writes them
    graph = {
        'sp_labels': sp_labels, 'sp_centroids': cent,
        'sp_length': rng.uniform(0.05, 3.0, (n_sp, 1)).astype(np.float32),
        'sp_surface': rng.uniform(0.01, 2.0, (n_sp, 1)).astype(np.float32),
        'sp_volume': rng.uniform(0.001, 1.0, (n_sp, 1)).astype(np.float32),
        'sp_point_count': counts[:, None].astype(np.uint64),
        'source': edges[:, :1].copy(), 'target': edges[:, 1:].copy(),
        'se_delta_mean': rng.normal(0.0, 1.0, (len(edges), 3)).astype(np.float32),
        'se_delta_std': rng.uniform(0.0, 1.0, (len(edges), 3)).astype(np.float32),
    }
    points = {}
    for i in range(n_sp):
        n = int(counts[i])
        P = np.empty((n, 15), dtype=np.float32)
        P[:, :3] = cent[i] + rng.normal(0.0, 0.3, (n, 3))
        P[:, 3:11] = rng.uniform(-0.5, 0.5, (n, 8))
        P[:, 11:14] = rng.uniform(0.0, 1.0, (n, 3))
        P[:, 14] = rng.normal(0.0, 1.0, n)
        points[i] = P
    return name, graph, points


def make_dataset(seed=0):
    """-> (train scenes, test scenes), each scene = (name, graph-file dict, {superpoint id: rows})."""
    rng = np.random.default_rng(4321 + seed)
    train = [make_scene(rng, 'train/room_%d' % i, int(rng.integers(40, 70))) for i in range(4)]
    test = [make_scene(rng, 'test/room_%d' % i, int(rng.integers(40, 70))) for i in range(2)]
    return train, test
