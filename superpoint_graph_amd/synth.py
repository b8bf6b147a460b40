"""Deterministic synthetic S3DIS-shaped superpoint graphs.

A scene is what `learning/spg.py:loader` (reference, :130-171) hands to `eccpc_collate`:
per-superpoint dense clouds [Nv, F, 128] (already resampled / normalised on the host, spg.py:209-222),
the diameter "global" feature, the validity flag (spg.py:203-204), a directed superedge list listed in
both directions (partition/graphs.py:105-108) with 13 standardised edge features (spg.py:51-64), and the
`targets` matrix (spg.py:71-73,108).  No dataset is available offline, so benchmarks and parity tests
use these seeded scenes; shapes follow BASELINE.json (1000 superpoints x 128 points x 14 features,
5000 directed superedges).
"""
from __future__ import annotations

import numpy as np


def scene(seed: int = 0, n_sp: int = 1000, n_edges: int = 5000, n_feat: int = 14, n_pts: int = 128,
          n_edge_feat: int = 13, n_classes: int = 13, minpts: int = 40, knn: int = 8,
          small_frac: float = 0.0):
    """Returns a dict with the per-scene sample (numpy):
      edges i64[E,2] (source,target) in "igraph edge order", edge_feats f32[E,Fe], n_sp,
      flag i64[N] (0 valid / -1 too small), clouds f32[Nv,F,P], diam f32[Nv],
      targets i64[N, 2+C] (col0 majority label or -100, col1 #unlabelled, col2.. class histogram).
    """
    rng = np.random.default_rng(1234 + seed)
    N = n_sp
    # raw point counts: lognormal around 600 points, capped like s3dis_dataset.py:155
    n_raw = np.clip(np.round(rng.lognormal(np.log(600.0), 1.0, N)), 1, 10000).astype(np.int64)
    if small_frac > 0:                      # force some too-small superpoints (tests / fixtures)
        n_raw[rng.random(N) < small_frac] = max(1, minpts // 4)
    flag = np.where(n_raw < minpts, -1, 0).astype(np.int64)
    nv = int((flag == 0).sum())

    clouds = np.empty((nv, n_feat, n_pts), dtype=np.float32)
    xyz = rng.uniform(-1.0, 1.0, (nv, 3, n_pts))
    xyz -= xyz.mean(2, keepdims=True)
    ext = (xyz.max(2) - xyz.min(2)).max(1)
    xyz /= (ext[:, None, None] + 1e-10)
    clouds[:, :3] = xyz
    if n_feat > 3:
        rest = rng.uniform(-0.5, 0.5, (nv, n_feat - 3, n_pts))
        if n_feat >= 14:
            rest[:, 8:11] = rng.uniform(0.0, 1.0, (nv, 3, n_pts))   # room-normalised XYZ in [0,1]
        clouds[:, 3:] = rest
    diam = rng.uniform(0.1, 3.0, nv).astype(np.float32)

    # superedges: each unordered pair joins a node with one of its `knn` nearest synthetic centroids
    cent = rng.uniform(0.0, 10.0, (N, 3))
    n_pairs = n_edges // 2
    d2 = ((cent[:, None, :] - cent[None, :, :]) ** 2).sum(-1) if N <= 4096 else None
    pairs = set()
    if d2 is not None:
        np.fill_diagonal(d2, np.inf)
        nbr = np.argsort(d2, axis=1)[:, :knn]
    else:  # large graphs: grid-free approximation, neighbours among random candidates
        nbr = None
    guard = 0
    while len(pairs) < n_pairs and guard < 50 * n_pairs:
        guard += 1
        i = int(rng.integers(N))
        if nbr is not None:
            j = int(nbr[i, int(rng.integers(min(knn, N - 1)))])
        else:
            cand = rng.integers(0, N, 4 * knn)
            cand = cand[cand != i]
            j = int(cand[np.argmin(((cent[cand] - cent[i]) ** 2).sum(1))])
        if i != j:
            pairs.add((min(i, j), max(i, j)))
    pairs = np.array(sorted(pairs), dtype=np.int64).reshape(-1, 2)
    perm = rng.permutation(len(pairs))
    pairs = pairs[perm]
    edges = np.concatenate([pairs, pairs[:, ::-1]], 0)           # both directions (graphs.py:105-108)
    edge_feats = rng.standard_normal((len(edges), n_edge_feat)).astype(np.float32)

    labels = rng.integers(0, n_classes, N).astype(np.int64)
    unl = rng.random(N) < 0.02
    targets = np.zeros((N, 2 + n_classes), dtype=np.int64)
    targets[np.arange(N), 2 + labels] = n_raw
    targets[:, 0] = labels
    targets[unl, 0] = -100
    targets[unl, 1] = n_raw[unl]
    targets[unl, 2:] = 0
    return dict(edges=edges, edge_feats=edge_feats, n_sp=N, flag=flag, clouds=clouds, diam=diam,
                targets=targets)


def collate_numpy(scenes):
    """The tensor side of `spg.eccpc_collate` (reference learning/spg.py:178-193) for synthetic
    scenes: concatenation with node offsets.  Index construction itself lives in
    `learning.ecc.GraphConvInfo`."""
    return dict(
        edge_lists=[s['edges'] for s in scenes],
        vcounts=[s['n_sp'] for s in scenes],
        edge_feats=[s['edge_feats'] for s in scenes],
        clouds_flag=np.concatenate([s['flag'] for s in scenes]),
        clouds=np.concatenate([s['clouds'] for s in scenes]),
        clouds_global=np.concatenate([s['diam'] for s in scenes]),
        targets=np.concatenate([s['targets'] for s in scenes]),
    )
