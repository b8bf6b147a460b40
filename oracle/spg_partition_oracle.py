"""TEST INFRASTRUCTURE (checker only -- nothing under superpoint_graph_amd/ imports this).

CPU restatements for the superpoint-graph construction row (SURVEY.md section 8, f4 tail):

* `sp_graph_after_triangulation`: the reference's partition/graphs.py:75-210 `compute_sp_graph` from the tetrahedra onwards,
  in numpy with the reference's own dtypes and operation order (float32 means / standard deviations, float64 covariance and
  eigenvalues).  PINNED: oracle/validate_against_reference.py runs the imported reference function on the same inputs and
  requires every array equal (integers) / equal to float32 round-off given the unspecified edge order (floats); the same run
  writes tests/golden/sp_graph.npz.
* `prune`: partition/ply_c/ply_c.cpp:288-382 in numpy, float32 in the reference's operation order.  PARITY UNPINNED (see `geof`).
* `geof`: partition/ply_c/ply_c.cpp:384-462 `compute_geof` in float64 numpy.  PARITY UNPINNED: the reference is a C++
  extension that needs Eigen and Boost.Python, neither of which is in this image, so it cannot be compiled (oracle/_ref) or
  imported; the restatement follows the published formulas (covariance of the k_nn + 1 neighbourhood, sorted eigenvalues,
  linearity / planarity / scattering / verticality) and is the only comparator the device kernel has.
"""
import numpy as np

_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))      # graphs.py:85-101


def interface_edges(tets, in_component, xyz, d_max):
    """graphs.py:85-113 -> int64 [2, n_edg]: unique directed vertex pairs joining two components, lexicographic order."""
    tets = np.asarray(tets, dtype=np.int64)
    comp = np.asarray(in_component)
    cols = []
    for a, b in _PAIRS:
        m = comp[tets[:, a]] != comp[tets[:, b]]
        cols.append(np.stack((tets[m, a], tets[m, b])))
        cols.append(np.stack((tets[m, b], tets[m, a])))
    edges = np.unique(np.concatenate(cols, axis=1), axis=1) if cols else np.zeros((2, 0), dtype=np.int64)      # :108
    if d_max > 0 and edges.shape[1]:
        dist = np.sqrt(((xyz[edges[0]] - xyz[edges[1]]) ** 2).sum(1))      # :111, float32 like the reference
        edges = edges[:, dist < d_max]
    return edges


def sp_graph_after_triangulation(xyz, d_max, in_component, components, labels, n_labels, tets):
    in_component = np.asarray(in_component)
    n_com = int(in_component.max()) + 1
    labels = np.asarray(labels)
    has_labels = len(labels) > 1
    label_hist = has_labels and labels.ndim > 1 and labels.shape[1] > 1
    edges = interface_edges(tets, in_component, xyz, d_max)
    ec = in_component[edges]
    index = n_com * ec[0].astype(np.int64) + ec[1]                        # :120
    order = np.argsort(index, kind='stable')                               # :121 (the reference's order inside a group is unspecified)
    edges, ec, index = edges[:, order], ec[:, order], index[order]
    starts = np.flatnonzero(np.r_[True, index[1:] != index[:-1]]) if index.size else np.zeros(0, dtype=np.int64)
    bounds = np.r_[starts, index.size]
    n_sedg = len(starts)
    g = {'is_nn': False}
    g['sp_centroids'] = np.zeros((n_com, 3), np.float32)
    for k in ('sp_length', 'sp_surface', 'sp_volume'):
        g[k] = np.zeros((n_com, 1), np.float32)
    g['sp_point_count'] = np.zeros((n_com, 1), np.uint64)
    g['source'] = np.zeros((n_sedg, 1), np.uint32)
    g['target'] = np.zeros((n_sedg, 1), np.uint32)
    for k, w in (('se_delta_mean', 3), ('se_delta_std', 3), ('se_delta_norm', 1), ('se_delta_centroid', 3), ('se_length_ratio', 1),
                 ('se_surface_ratio', 1), ('se_volume_ratio', 1), ('se_point_count_ratio', 1)):
        g[k] = np.zeros((n_sedg, w), np.float32)
    g['sp_labels'] = np.zeros((n_com, n_labels + 1), np.uint32) if has_labels else []
    for c in range(n_com):                                                 # :141-172
        comp = components[c]
        if has_labels and not label_hist:
            g['sp_labels'][c] = np.histogram(labels[comp], bins=np.arange(n_labels + 2) - 0.5)[0]
        if label_hist:
            g['sp_labels'][c] = labels[comp].sum(0)
        g['sp_point_count'][c] = len(comp)
        pts = np.unique(xyz[comp], axis=0)
        if len(pts) == 1:
            g['sp_centroids'][c] = pts
        elif len(pts) == 2:
            g['sp_centroids'][c] = pts.mean(0)
            g['sp_length'][c] = np.sqrt(np.sum(np.var(pts, axis=0)))
        else:
            ev = -np.sort(-np.linalg.eig(np.cov(pts.T, rowvar=True))[0].real)
            g['sp_centroids'][c] = pts.mean(0)
            g['sp_length'][c] = ev[0]
            g['sp_surface'][c] = np.sqrt(ev[0] * ev[1] + 1e-10)
            g['sp_volume'][c] = np.sqrt(ev[0] * ev[1] * ev[2] + 1e-10)
    for s in range(n_sedg):                                                # :174-208
        sl = slice(bounds[s], bounds[s + 1])
        cs, ct = ec[0, bounds[s]], ec[1, bounds[s]]
        g['source'][s], g['target'][s] = cs, ct
        g['se_delta_centroid'][s] = g['sp_centroids'][cs] - g['sp_centroids'][ct]
        g['se_length_ratio'][s] = g['sp_length'][cs] / (g['sp_length'][ct] + 1e-6)
        g['se_surface_ratio'][s] = g['sp_surface'][cs] / (g['sp_surface'][ct] + 1e-6)
        g['se_volume_ratio'][s] = g['sp_volume'][cs] / (g['sp_volume'][ct] + 1e-6)
        g['se_point_count_ratio'][s] = g['sp_point_count'][cs] / (g['sp_point_count'][ct] + 1e-6)
        delta = xyz[edges[0, sl]] - xyz[edges[1, sl]]
        if len(delta) > 1:
            g['se_delta_mean'][s] = delta.mean(0)
            g['se_delta_std'][s] = delta.std(0)
            g['se_delta_norm'][s] = np.mean(np.sqrt(np.sum(delta ** 2, axis=1)))
        else:
            g['se_delta_mean'][s] = delta
            g['se_delta_norm'][s] = np.sqrt(np.sum(delta ** 2))
    return g


def geof(xyz, target, k_nn):
    """ply_c.cpp:384-462 in float64 -> float32 [n,4] (linearity, planarity, scattering, verticality)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    n = xyz.shape[0]
    nb = np.concatenate((np.arange(n)[:, None], np.asarray(target, dtype=np.int64).reshape(n, k_nn)), axis=1)
    pos = xyz[nb]                                                          # [n, k+1, 3]  (:398-412)
    cen = pos - pos.mean(1, keepdims=True)
    cov = np.einsum('nki,nkj->nij', cen, cen) / (k_nn + 1)                  # :414-415
    w, v = np.linalg.eigh(cov)                                             # ascending
    w, v = w[:, ::-1], v[:, :, ::-1]                                       # descending, eigenvectors in columns (:420-436)
    lam = np.maximum(w, 0.0)
    s = np.sqrt(lam)
    with np.errstate(divide='ignore', invalid='ignore'):
        lin = (s[:, 0] - s[:, 1]) / s[:, 0]
        pla = (s[:, 1] - s[:, 2]) / s[:, 0]
        sca = s[:, 2] / s[:, 0]
        u = np.einsum('nk,ndk->nd', lam, np.abs(v))                        # :441-444
        ver = u[:, 2] / np.sqrt((u ** 2).sum(1))
    return np.stack((lin, pla, sca, ver), 1).astype(np.float32)


def prune(xyz, voxel_size, rgb, labels, objects, n_labels, n_objects):
    """ply_c.cpp:288-382 in numpy with the reference's float32 operations in the reference's order (np.add.at is unbuffered: the
    additions of one voxel happen in input order, like the serial loop).  PARITY UNPINNED like `geof` (Boost.Python extension)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    vs = np.float32(voxel_size)
    bins = np.floor((xyz - xyz.min(0)) / vs).astype(np.uint32)                       # :326-328
    _, first, inv = np.unique(bins, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind='stable')] = np.arange(len(first))                    # voxel index = order of first occurrence (:166-176)
    vox = rank[inv]
    V = len(first)
    acc = np.zeros((V, 3), np.float32)
    np.add.at(acc, vox, xyz)                                                          # :259-261, float32, input order
    col = np.zeros((V, 3), np.uint32)
    np.add.at(col, vox, np.asarray(rgb, dtype=np.uint32).reshape(-1, 3))
    count = np.bincount(vox, minlength=V).astype(np.float32)
    out_xyz = acc / count[:, None]                                                    # :365-368
    out_rgb = (col.astype(np.float32) / count[:, None]).astype(np.uint8)              # :371-374 (truncation)
    lab = np.zeros((V, n_labels + 1), np.uint32)
    obj = np.zeros((V, n_objects + 1), np.uint32)
    if n_labels > 0:
        np.add.at(lab, (vox, np.asarray(labels, dtype=np.int64).reshape(-1)), 1)
        if n_objects > 0:
            np.add.at(obj, (vox, np.asarray(objects, dtype=np.int64).reshape(-1)), 1)
    return out_xyz, out_rgb, lab, obj


def synthetic_cloud(seed, n=3000, n_blobs=24, duplicates=20):
    """A small labelled cloud with a partition: blobs of different shapes (lines, planes, balls -> all branches of :151-172),
    one single-point and one two-point component, and a few exactly duplicated points (np.unique(xyz[comp], axis=0))."""
    rng = np.random.default_rng(4321 + seed)
    centers = rng.uniform(-4, 4, (n_blobs, 3))
    which = rng.integers(0, n_blobs, n)
    scale = rng.uniform(0.02, 0.6, (n_blobs, 3))
    scale[::3, 1:] *= 0.02                       # line-like blobs
    scale[1::3, 2] *= 0.02                       # plane-like blobs
    xyz = (centers[which] + rng.normal(size=(n, 3)) * scale[which]).astype(np.float32)
    comp = which.copy()
    extra = np.array([[9, 9, 9], [-9, 9, 9], [-9.5, 9, 9.25]], dtype=np.float32)      # components n_blobs (1 point), n_blobs + 1 (2 points)
    xyz = np.concatenate((xyz, extra))
    comp = np.concatenate((comp, [n_blobs, n_blobs + 1, n_blobs + 1]))
    dup = rng.integers(0, n, duplicates)         # exact duplicates inside their component
    xyz = np.concatenate((xyz, xyz[dup]))
    comp = np.concatenate((comp, comp[dup]))
    _, comp = np.unique(comp, return_inverse=True)
    perm = rng.permutation(len(xyz))
    xyz, comp = np.ascontiguousarray(xyz[perm]), comp[perm].astype(np.int64)
    labels = rng.integers(0, 6, len(xyz))
    labels[rng.random(len(xyz)) < 0.1] = 0
    components = [np.flatnonzero(comp == c) for c in range(comp.max() + 1)]
    return xyz, comp, components, labels
