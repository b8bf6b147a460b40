"""`compute_sp_graph` / `compute_geof` with the reference's signatures (partition/graphs.py:75, partition/ply_c/ply_c.cpp:384 as
bound by `libply_c.compute_geof(xyz, target, k_nn)`), computed by the HIP library (csrc/spg_spgraph.hip).

What stays on the host: scipy's Delaunay triangulation (qhull; a different algorithm class) -- everything
after it (12 T candidate pairs -> unique interface edges -> d_max filter -> ordering by component pair -> per-superpoint and
per-superedge features, the part the reference does with np.unique over 2 x 12T columns and two Python loops) runs on the GPU.
Integer outputs are bit-identical to the reference; float features are accumulated in float64 (the reference: float32) and
agree to float32 round-off (tests/test_gpu_spgraph.py).  No CPU fallback."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('superpoint_graph_amd.partition has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def compute_sp_graph(xyz, d_max, in_component, components, labels, n_labels, tetrahedra=None):
    """compute the superpoint graph with superpoints and superedges features (reference partition/graphs.py:75-210).
    xyz float32 [n,3]; in_component [n]; components: list of index arrays (only its length is used: in_component carries the
    same information); labels: [] / [n] integer labels / [n, n_labels + 1] label histograms.
    tetrahedra (optional): int [T,4] simplices of an already computed triangulation (default: scipy.spatial.Delaunay(xyz))."""
    dev = _dev()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    in_component = np.asarray(in_component)
    if xyz.ndim != 2 or xyz.shape[1] != 3 or in_component.shape != (len(xyz),):
        raise ValueError('compute_sp_graph: xyz [n,3] and in_component [n] expected')
    if len(xyz) == 0 or int(in_component.min()) < 0:
        raise ValueError('compute_sp_graph: empty cloud or negative component index')
    n_com = int(in_component.max()) + 1
    if components is not None and len(components) != n_com:
        raise ValueError(f'compute_sp_graph: {len(components)} components for max(in_component) + 1 = {n_com}')
    labels = np.asarray(labels)
    has_labels = len(labels) > 1
    label_hist = has_labels and labels.ndim > 1 and labels.shape[1] > 1
    if tetrahedra is None:
        from scipy.spatial import Delaunay
        tetrahedra = Delaunay(xyz).simplices          # (`tri.vertices` in the reference: the attribute's old name)
    tetrahedra = np.ascontiguousarray(tetrahedra, dtype=np.int32).reshape(-1, 4)
    if tetrahedra.size and (int(tetrahedra.min()) < 0 or int(tetrahedra.max()) >= len(xyz)):
        raise IndexError('compute_sp_graph: a tetrahedron vertex is outside [0, number of points)')
    tets = ops.upload(torch.from_numpy(tetrahedra), dev)
    xyz_d = ops.upload(torch.from_numpy(xyz), dev)
    comp_d = ops.upload(torch.from_numpy(np.ascontiguousarray(in_component, dtype=np.int32)), dev)
    lab_d = rows_d = None
    if has_labels and not label_hist:
        lab = labels.reshape(-1)
        if not np.issubdtype(lab.dtype, np.integer):
            if not np.all(lab == np.rint(lab)):
                raise NotImplementedError('compute_sp_graph: non-integer point labels')
        lab_d = ops.upload(torch.from_numpy(np.ascontiguousarray(lab, dtype=np.int32)), dev)
    elif label_hist:
        if labels.shape[1] != n_labels + 1:
            raise ValueError(f'compute_sp_graph: label histograms need {n_labels + 1} columns, got {labels.shape[1]}')
        rows_d = ops.upload(torch.from_numpy(np.ascontiguousarray(labels).astype(np.uint32).view(np.int32)), dev)
    g = ops.sp_graph(xyz_d, comp_d, n_com, tets, float(d_max), lab_d, rows_d, int(n_labels))

    def host(k, dtype):
        return g[k].cpu().numpy().view(dtype) if g[k].dtype != torch.float32 else g[k].cpu().numpy()
    graph = dict([("is_nn", False)])
    graph["sp_centroids"] = host('sp_centroids', np.float32)
    graph["sp_length"] = host('sp_length', np.float32)
    graph["sp_surface"] = host('sp_surface', np.float32)
    graph["sp_volume"] = host('sp_volume', np.float32)
    graph["sp_point_count"] = host('sp_point_count', np.uint64)
    graph["source"] = host('source', np.uint32)
    graph["target"] = host('target', np.uint32)
    for k in ('se_delta_mean', 'se_delta_std', 'se_delta_norm', 'se_delta_centroid', 'se_length_ratio', 'se_surface_ratio',
              'se_volume_ratio', 'se_point_count_ratio'):
        graph[k] = host(k, np.float32)
    graph["sp_labels"] = host('sp_labels', np.uint32) if has_labels else []
    return graph


def compute_geof(xyz, target, k_nn):
    """linearity, planarity, scattering, verticality of every point from its k_nn nearest neighbours (reference
    partition/ply_c/ply_c.cpp:384-462, called as libply_c.compute_geof(xyz, graph_nn['target'], k_nn)) -> float32 [n,4]."""
    dev = _dev()
    xyz_d = ops.upload(torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32)), dev)
    tgt = np.ascontiguousarray(target).reshape(-1).astype(np.uint32)
    if tgt.size and int(tgt.max()) >= len(xyz):
        raise IndexError('compute_geof: a neighbour index is outside [0, number of points)')
    tgt = tgt.view(np.int32)
    return ops.compute_geof(xyz_d, ops.upload(torch.from_numpy(tgt), dev), int(k_nn)).cpu().numpy()


def prune(xyz, voxel_size, rgb, labels, objects, n_labels, n_objects):
    """prune the point cloud xyz with a regular voxel grid (reference partition/ply_c/ply_c.cpp:288-382, called as
    libply_c.prune(xyz, voxel_width, rgb, labels, objects, n_labels, n_objects)) -> (xyz float32 [V,3], rgb uint8 [V,3],
    labels uint32 [V, n_labels+1], objects uint32 [V, n_objects+1]).  Like the reference, labels are read only when
    n_labels > 0 and objects only when n_objects > 0 (callers pass dummy arrays otherwise)."""
    dev = _dev()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = len(xyz)
    up = lambda a: ops.upload(torch.from_numpy(a), dev)
    rgb_d = up(np.ascontiguousarray(rgb, dtype=np.uint8).reshape(n, 3))
    lab_d = up(np.ascontiguousarray(labels, dtype=np.uint8).reshape(n)) if n_labels > 0 else None
    obj_d = up(np.ascontiguousarray(objects, dtype=np.uint32).reshape(n).view(np.int32)) if (n_labels > 0 and n_objects > 0) else None
    x, c, l, o = ops.prune(up(xyz), float(np.float32(voxel_size)), rgb_d, lab_d, obj_d, int(n_labels), int(n_objects))
    return x.cpu().numpy(), c.cpu().numpy(), l.cpu().numpy().view(np.uint32), o.cpu().numpy().view(np.uint32)
