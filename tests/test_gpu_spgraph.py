"""GPU: superpoint-graph construction (csrc/spg_spgraph.hip through superpoint_graph_amd/partition/graphs.py) --
SURVEY.md section 8, row f4 tail.

* against the golden outputs of the imported REFERENCE function partition/graphs.py:compute_sp_graph (tests/golden/sp_graph.npz):
  every integer output bit-exact (the edge set after the float32 d_max filter, source / target, point counts, label
  histograms); float features within the stated tolerances (the device accumulates in float64 and rounds once, the reference
  accumulates in float32; the order of the Delaunay edges inside a superedge is unspecified in the reference);
* against the CPU oracle on another cloud, with the product computing its own triangulation;
* at partition scale (300 000 points, ~1.9 M tetrahedra) through size-independent properties;
* compute_geof against the float64 restatement (parity unpinned: ply_c is a C++ extension that cannot be built here)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import spg_partition_oracle as P
from test_partition_oracle import golden_case

pytestmark = pytest.mark.gpu

# max |device - reference| relative to max |reference| of the array
TOL = {'sp_centroids': 2e-6,            # float32 running mean of up to ~200 points in the reference vs float64 here
       'sp_length': 2e-6, 'sp_surface': 2e-6, 'sp_volume': 2e-6,       # same float64 covariance; Jacobi vs LAPACK geev
       'se_delta_mean': 2e-6, 'se_delta_std': 2e-6, 'se_delta_norm': 2e-6,
       'se_delta_centroid': 1e-5,       # difference of two centroids
       'se_length_ratio': 1e-5, 'se_surface_ratio': 1e-5, 'se_volume_ratio': 1e-5, 'se_point_count_ratio': 0.0}


def compare(mine, ref, where):
    for k, a in ref.items():
        b = mine[k]
        assert a.shape == b.shape and a.dtype == b.dtype, (where, k, a.shape, b.shape, a.dtype, b.dtype)
        if a.dtype.kind in 'ui':
            assert np.array_equal(a, b), f'{where}: {k} (integer) differs'
        else:
            err = np.abs(a.astype(np.float64) - b).max() / max(float(np.abs(a).max()), 1e-30) if a.size else 0.0
            assert err <= TOL[k], f'{where}: {k} {err:.2e} > {TOL[k]:.0e}'


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_sp_graph_vs_reference_golden(hip, tag):
    from superpoint_graph_amd.partition import graphs
    g = np.load(os.path.join(GOLDEN, 'sp_graph.npz'))
    (xyz, d_max, comp, components, labels, n_labels, tets), ref = golden_case(g, tag)
    mine = graphs.compute_sp_graph(xyz, d_max, comp, components, labels, n_labels, tetrahedra=tets)
    assert mine['is_nn'] is False
    if n_labels == 0:
        assert mine['sp_labels'] == []
    compare({k: v for k, v in mine.items() if k in ref}, ref, f'golden {tag}')
    again = graphs.compute_sp_graph(xyz, d_max, comp, components, labels, n_labels, tetrahedra=tets)
    for k in ref:                                           # deterministic: no order-dependent float atomics anywhere
        assert np.array_equal(mine[k], again[k]), k


def test_sp_graph_own_triangulation_and_label_histograms_vs_oracle(hip):
    from scipy.spatial import Delaunay
    from superpoint_graph_amd.partition import graphs
    xyz, comp, components, labels = P.synthetic_cloud(7, n=20000, n_blobs=150, duplicates=200)
    hist = np.zeros((len(xyz), 7), dtype=np.uint32)
    hist[np.arange(len(xyz)), labels] = 1
    hist[:, 6] = 2
    mine = graphs.compute_sp_graph(xyz, 0.9, comp, components, hist, 6)
    ref = P.sp_graph_after_triangulation(xyz, 0.9, comp, components, hist, 6, Delaunay(xyz).simplices)
    assert len(ref['source']) > 300
    compare({k: v for k, v in mine.items() if k in ref and k != 'is_nn'}, {k: v for k, v in ref.items() if k != 'is_nn'}, 'oracle')


def test_sp_graph_degenerate_inputs(hip):
    from superpoint_graph_amd.partition import graphs
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(500, 3)).astype(np.float32)
    one = graphs.compute_sp_graph(xyz, 0.0, np.zeros(500, dtype=np.int64), [np.arange(500)], [], 0)      # a single component: no superedge
    assert one['source'].shape == (0, 1) and one['se_delta_mean'].shape == (0, 3) and one['sp_point_count'][0, 0] == 500
    comp = (xyz[:, 0] > 0).astype(np.int64)
    far = graphs.compute_sp_graph(xyz, 1e-6, comp, [np.flatnonzero(comp == 0), np.flatnonzero(comp == 1)], [], 0)   # d_max removes every edge
    assert far['source'].shape == (0, 1)
    two = graphs.compute_sp_graph(xyz, 0.0, comp, [np.flatnonzero(comp == 0), np.flatnonzero(comp == 1)], [], 0)
    assert two['source'][:, 0].tolist() == [0, 1] and two['target'][:, 0].tolist() == [1, 0]
    assert np.array_equal(two['se_delta_mean'][0], -two['se_delta_mean'][1])


def test_sp_graph_at_partition_scale_properties(hip):
    """300 000 points in 2 000 components (~1.9 M tetrahedra): properties that do not need the Python-loop oracle."""
    from scipy.spatial import Delaunay
    from superpoint_graph_amd import ops
    rng = np.random.default_rng(11)
    n, n_com = 300_000, 2000
    centers = rng.uniform(-20, 20, (n_com, 3))
    which = rng.integers(0, n_com, n)
    xyz = (centers[which] + rng.normal(size=(n, 3)) * 0.4).astype(np.float32)
    _, comp = np.unique(which, return_inverse=True)
    n_com = int(comp.max()) + 1
    tets = Delaunay(xyz).simplices.astype(np.int32)
    dev = torch.device('cuda')
    xyz_d, comp_d, tets_d = torch.from_numpy(xyz).to(dev), torch.from_numpy(comp.astype(np.int32)).to(dev), torch.from_numpy(tets).to(dev)
    labels = torch.from_numpy(rng.integers(0, 9, n).astype(np.int32)).to(dev)
    g = ops.sp_graph(xyz_d, comp_d, n_com, tets_d, 1.0, labels=labels, n_labels=8)
    src, tgt = g['source'][:, 0].cpu().numpy().astype(np.int64), g['target'][:, 0].cpu().numpy().astype(np.int64)
    key = src * n_com + tgt
    assert (np.diff(key) > 0).all() and (src != tgt).all()                                  # ordered by component pair, one row each, no self edge
    rev = np.searchsorted(key, tgt * n_com + src)
    assert np.array_equal(key[rev], tgt * n_com + src)                                     # every superedge has its reverse
    dm = g['se_delta_mean'].cpu().numpy()
    assert np.array_equal(dm, -dm[rev])                                                    # ... with exactly negated mean offset
    assert np.array_equal(g['se_delta_std'].cpu().numpy(), g['se_delta_std'].cpu().numpy()[rev])
    assert np.array_equal(g['se_delta_norm'].cpu().numpy(), g['se_delta_norm'].cpu().numpy()[rev])
    # the Delaunay edges behind the superedges: unique, interface only, shorter than d_max (float32 like the reference)
    e = g['edges'].cpu().numpy().view(np.uint64)
    a, b = (e >> np.uint64(32)).astype(np.int64), (e & np.uint64(0xffffffff)).astype(np.int64)
    assert len(np.unique(e)) == len(e) and (comp[a] != comp[b]).all()
    assert (np.sqrt(((xyz[a] - xyz[b]) ** 2).sum(1)) < np.float32(1.0)).all()
    off = g['seg_off'].cpu().numpy()
    assert off[0] == 0 and off[-1] == len(e) and (np.diff(off) > 0).all()
    assert np.array_equal(comp[a[off[:-1]]], src) and np.array_equal(comp[b[off[:-1]]], tgt)
    # exact integer accounting
    assert np.array_equal(g['sp_point_count'][:, 0].cpu().numpy(), np.bincount(comp, minlength=n_com))
    hist = g['sp_labels'].cpu().numpy()
    assert hist.sum() == n and np.array_equal(hist.sum(1), np.bincount(comp, minlength=n_com))
    # float64 reference values of the component means / eigenvalue identities
    cen = g['sp_centroids'].cpu().numpy().astype(np.float64)
    want = np.stack([np.bincount(comp, xyz[:, d].astype(np.float64), n_com) for d in range(3)], 1) / np.bincount(comp, minlength=n_com)[:, None]
    assert np.abs(cen - want).max() < 5e-6                                                  # (no duplicated points in this cloud)
    ev0 = g['sp_length'][:, 0].cpu().numpy().astype(np.float64)
    c0 = 5
    pts = xyz[comp == c0].astype(np.float64)
    assert abs(ev0[c0] - np.linalg.eigvalsh(np.cov(pts.T))[-1]) < 1e-6 * ev0[c0]


def test_compute_geof_vs_restatement(hip):
    from scipy.spatial import cKDTree
    from superpoint_graph_amd.partition import graphs
    rng = np.random.default_rng(3)
    n, k = 50_000, 45
    xyz = np.concatenate((rng.normal(size=(n // 2, 3)) * [4, 4, 0.05], rng.normal(size=(n - n // 2, 3)) * [0.05, 3, 3] + [8, 0, 0])).astype(np.float32)
    _, nb = cKDTree(xyz).query(xyz, k + 1)
    target = nb[:, 1:].astype(np.uint32).reshape(-1)
    mine = graphs.compute_geof(xyz, target, k)
    ref = P.geof(xyz, target, k)
    assert mine.shape == (n, 4) and mine.dtype == np.float32
    assert np.abs(mine - ref).max() < 2e-5                      # both float64 inside; float32 results
    assert (mine[: n // 2, 3] < 0.2).mean() > 0.95              # horizontal slab: low verticality
    assert (mine[n // 2:, 3] > 0.3).mean() > 0.9                # vertical slab


def _cloud(n, seed):
    rng = np.random.default_rng(seed)
    xyz = (rng.normal(size=(n, 3)) * [8, 6, 1.5] + [100, -50, 3]).astype(np.float32)
    xyz[rng.integers(0, n, n // 50)] = xyz[rng.integers(0, n, n // 50)]              # exact duplicates
    rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    labels = rng.integers(0, 9, n).astype(np.uint8)
    objects = rng.integers(0, 41, n).astype(np.uint32)
    return xyz, rgb, labels, objects


@pytest.mark.parametrize('n,voxel', [(200_000, 0.05), (50_000, 0.5), (3_000, 100.0)])
def test_prune_is_bit_exact_vs_restatement(hip, n, voxel):
    """Every output of the voxel-grid subsampling equal to the float32 restatement of ply_c.cpp:288-382, bit for bit: voxel
    order (first occurrence), positions (float32 sums in input order), truncated colour means, label / object histograms."""
    from superpoint_graph_amd.partition import libply_c
    xyz, rgb, labels, objects = _cloud(n, n)
    for n_labels, n_objects in ((8, 40), (8, 0), (0, 0)):
        got = libply_c.prune(xyz, voxel, rgb, labels, objects, n_labels, n_objects)
        ref = P.prune(xyz, voxel, rgb, labels, objects, n_labels, n_objects)
        for a, b, what in zip(got, ref, ('xyz', 'rgb', 'labels', 'objects')):
            assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
            assert np.array_equal(a, b), f'{what} differs (n_labels {n_labels}, n_objects {n_objects})'
        assert got[2].sum() == (n if n_labels else 0) and got[3].sum() == (n if n_objects and n_labels else 0)
    if voxel == 100.0:
        assert len(got[0]) == 1                                                        # one voxel: the mean of the whole cloud, serial float32 sum


def test_prune_known_answer_and_errors(hip):
    from superpoint_graph_amd.partition import libply_c
    xyz = np.array([[0, 0, 0], [0.9, 0.1, 0.2], [1.5, 0, 0], [0.2, 0.2, 0.2], [1.6, 0.1, 0.1]], np.float32)
    rgb = np.array([[10, 20, 30], [20, 30, 40], [100, 100, 100], [30, 40, 50], [101, 103, 105]], np.uint8)
    lab = np.array([1, 2, 0, 1, 2], np.uint8)
    x, c, l, o = libply_c.prune(xyz, 1.0, rgb, lab, np.zeros(1, dtype=np.uint8), 2, 0)
    assert c.tolist() == [[20, 30, 40], [100, 101, 102]] and l.tolist() == [[0, 2, 1], [1, 0, 1]] and o.tolist() == [[0], [0]]
    assert np.allclose(x, [[1.1 / 3, 0.1, 0.4 / 3], [1.55, 0.05, 0.05]], atol=1e-7)
    with pytest.raises(IndexError):
        libply_c.prune(xyz, 1.0, rgb, lab, np.zeros(1, dtype=np.uint8), 1, 0)         # label 2 with n_labels = 1 (vector::at throws in the reference)
    with pytest.raises(ValueError):
        libply_c.prune(xyz * 1e6, 1e-3, rgb, lab, np.zeros(1, dtype=np.uint8), 2, 0)  # > 2^21 bins along an axis
