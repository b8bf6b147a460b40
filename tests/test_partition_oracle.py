"""CPU: the superpoint-graph construction oracle (oracle/spg_partition_oracle.py) against the golden outputs of the imported
reference function partition/graphs.py:compute_sp_graph (tests/golden/sp_graph.npz, oracle/validate_against_reference.py), and
known-answer properties of the compute_geof restatement (ply_c.cpp:384-462; parity unpinned -- see the oracle's header)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import spg_partition_oracle as P

ORDER_FREE = ('sp_', 'source', 'target', 'se_delta_centroid', 'se_length_ratio', 'se_surface_ratio', 'se_volume_ratio',
              'se_point_count_ratio')


def golden_case(g, tag):
    comp = g[f'{tag}/comp']
    components = [np.flatnonzero(comp == c) for c in range(int(comp.max()) + 1)]
    ref = {k.split('/ref/')[1]: g[k] for k in g.files if k.startswith(f'{tag}/ref/')}
    return (g[f'{tag}/xyz'], float(g[f'{tag}/d_max']), comp, components, g[f'{tag}/labels'], int(g[f'{tag}/n_labels']),
            g[f'{tag}/tets']), ref


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_sp_graph_oracle_reproduces_the_reference(tag):
    g = np.load(os.path.join(GOLDEN, 'sp_graph.npz'))
    (xyz, d_max, comp, components, labels, n_labels, tets), ref = golden_case(g, tag)
    mine = P.sp_graph_after_triangulation(xyz, d_max, comp, components, labels, n_labels, tets)
    assert len(ref['source']) > 10
    for k, a in ref.items():
        b = mine[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if k.startswith(ORDER_FREE):
            assert np.array_equal(a, b), k                    # integers, superpoint features, ratios: bit-equal
        else:                                                 # float32 sums over a group whose edge order the reference leaves open
            assert np.abs(a.astype(np.float64) - b).max() <= 1e-6 * np.abs(a).max(), k


def test_sp_graph_golden_covers_every_branch():
    g = np.load(os.path.join(GOLDEN, 'sp_graph.npz'))
    (xyz, d_max, comp, components, labels, n_labels, tets), ref = golden_case(g, 'a')
    uniq = [len(np.unique(xyz[c], axis=0)) for c in components]
    assert 1 in uniq and 2 in uniq and max(uniq) > 100                       # graphs.py:151 / :156 / :161
    assert any(len(c) > u for c, u in zip(components, uniq))                  # duplicated points inside a component
    assert ref['sp_labels'].sum() == len(xyz)                                # every label in range is counted once
    sizes = np.diff(np.flatnonzero(np.r_[True, np.diff(ref['source'][:, 0].astype(np.int64) * 1000 + ref['target'][:, 0]) != 0, True]))
    assert (sizes == 1).all()                                                # one row per (source, target) pair


def test_geof_known_answers():
    rng = np.random.default_rng(5)
    n, k = 64, 12
    t = np.linspace(-1, 1, n)
    line = np.stack((t, 2 * t, 0.5 * t), 1) + rng.normal(size=(n, 3)) * 1e-6
    nbr = np.array([[(i + d) % n for d in range(1, k + 1)] for i in range(n)])
    f = P.geof(line.astype(np.float32), nbr, k)
    assert (f[:, 0] > 0.999).all() and (f[:, 1] < 1e-3).all() and (f[:, 2] < 1e-3).all()          # linear
    plane = np.stack((rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.zeros(n)), 1)
    f = P.geof(plane.astype(np.float32), nbr, k)
    assert (f[:, 2] < 1e-6).all() and (f[:, 0] + f[:, 1] > 0.999).all()                            # planar: no scattering
    assert (f[:, 3] < 1e-6).all()                                                                   # horizontal plane: verticality 0
    ball = rng.normal(size=(4000, 3))
    nb = np.array([rng.choice(4000, 300, replace=False) for _ in range(16)])
    f = P.geof(ball.astype(np.float32), np.concatenate([nb, np.zeros((4000 - 16, 300), dtype=np.int64)]), 300)[:16]
    assert (f[:, 2] > 0.75).all()                                                                   # isotropic: scattering near 1
    assert np.allclose(f[:, 0] + f[:, 1] + f[:, 2], 1.0, atol=1e-6)


def test_prune_known_answer():
    """Five points, two voxels: hand-computed means, truncated colour means, label histograms, voxel order = first occurrence."""
    xyz = np.array([[0, 0, 0], [0.9, 0.1, 0.2], [1.5, 0, 0], [0.2, 0.2, 0.2], [1.6, 0.1, 0.1]], np.float32)
    rgb = np.array([[10, 20, 30], [20, 30, 40], [100, 100, 100], [30, 40, 50], [101, 103, 105]], np.uint8)
    lab = np.array([1, 2, 0, 1, 2], np.uint8)
    x, c, l, o = P.prune(xyz, 1.0, rgb, lab, np.zeros(1), 2, 0)
    assert np.allclose(x, [[1.1 / 3, 0.1, 0.4 / 3], [1.55, 0.05, 0.05]], atol=1e-7)
    assert c.tolist() == [[20, 30, 40], [100, 101, 102]] and l.tolist() == [[0, 2, 1], [1, 0, 1]] and o.tolist() == [[0], [0]]
    x2, _, l2, _ = P.prune(xyz[::-1].copy(), 1.0, rgb[::-1].copy(), lab[::-1].copy(), np.zeros(1), 2, 0)      # reversed input: voxel order flips
    assert np.allclose(x2, x[::-1], atol=1e-7) and l2.tolist() == l[::-1].tolist()
