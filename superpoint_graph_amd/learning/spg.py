"""Batching contract of the reference's learning/spg.py for the hot path: `eccpc_collate` turns a list of
loader samples `(targets, graph, clouds_meta, clouds_flag, clouds, clouds_global)` (reference
learning/spg.py:130-171) into `(targets, [GraphConvInfo], (clouds_meta, clouds_flag, clouds,
clouds_global))` (reference learning/spg.py:178-193), and `load_superpoints_device` is the device-side form of the
per-superpoint loader (`load_superpoint` + `augment_cloud`, :198-258) over a ragged point buffer.  File readers are out of scope;
`SuperpointGraph` is a minimal stand-in for the igraph.Graph API that GraphConvInfo.set_batch touches, so
synthetic scenes (and tests) need no igraph install.  Real igraph graphs work unchanged (duck typing)."""
import numpy as np
import torch

from . import ecc


class _EdgeSeq:
    def __init__(self, g, idx=None):
        self._g, self._idx = g, idx

    def __getitem__(self, idx):
        return _EdgeSeq(self._g, list(idx))

    def attributes(self):
        return list(self._g._eattrs.keys())

    def get_attribute_values(self, a):
        vals = self._g._eattrs[a]
        idx = range(len(vals)) if self._idx is None else self._idx
        return [vals[i] for i in idx]


class _Vertex:
    def __init__(self, g, i):
        self._g, self.index = g, i

    def __getitem__(self, a):
        return self._g._vattrs[a][self.index]


class _VertexSeq:
    def __init__(self, g):
        self._g = g

    def __len__(self):
        return self._g._n

    def __iter__(self):
        return (_Vertex(self._g, i) for i in range(self._g._n))

    def __getitem__(self, key):
        if isinstance(key, str):
            return list(self._g._vattrs[key])
        return _Vertex(self._g, int(key))


class SuperpointGraph:
    """Directed graph with edge / vertex attributes: the subset of the igraph.Graph API that the reference's loader and
    batching touch (learning/spg.py:109-143,151-166, learning/ecc/GraphConvInfo.py:48-58): `vcount`, `get_edgelist`,
    `es[...]`, `es.attributes`, `indegree`, `vs[...]`, `permute_vertices`, `neighborhood`, `subgraph`.

    The igraph semantics this class ASSUMES, method by method (python-igraph's documented behaviour; hand-written expected
    values in tests/test_host.py::test_superpoint_graph_igraph_semantics), and what the callers rely on:
      * `permute_vertices(perm)`: "vertex k of the original graph becomes vertex perm[k] in the new graph" (python-igraph
        Graph.permute_vertices); vertex attributes travel with their vertex, edges keep their order and attributes.  Used by
        the loader with a random permutation (spg.py:136-137): only the multiset of vertices matters afterwards.
      * `neighborhood(vertices, order)` (default mode='all'): for every given vertex the vertices reachable in at most
        `order` steps IGNORING edge direction, the vertex itself included.  igraph returns them in breadth-first order; the only
        caller turns the result into `sorted(set(...))` (random_neighborhoods, spg.py:118-121), so only MEMBERSHIP is
        observable -- this class returns the centre first and every further level sorted.
      * `subgraph(vertices)` (= induced_subgraph): the kept vertices are renumbered 0..k-1 in increasing order of their old
        ids (igraph keeps the original relative order whatever order the ids are passed in; both call sites pass increasing
        ids: a sorted set, spg.py:120-121, and range(n), spg.py:127); an edge survives iff both endpoints do; vertex and edge
        attributes follow.  Edge ORDER of the result: igraph's implementation-dependent ('copy_and_delete' keeps the original
        relative order, 'create_from_scratch' -- chosen automatically for small selections -- does not); this class keeps the
        original order.  Nothing downstream observes it: GraphConvInfo.set_batch re-orders the edges by target, the
        aggregation over a target's in-edges is a mean (order changes fp32 round-off only) and the edge features travel with
        their edges.
      * `indegree`, `vcount`, `get_edgelist`, `es.attributes()`, `es[idx].get_attribute_values(a)`, `vs[i][a]`, `vs[a]`:
        plain accessors (GraphConvInfo.py:48-58, spg.py:153-166)."""

    def __init__(self, n, edges, directed=True, edge_attrs=None, vertex_attrs=None):
        self._n = int(n)
        self._edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        self._eattrs = {k: list(v) for k, v in dict(edge_attrs or {}).items()}
        self._vattrs = {k: list(v) for k, v in dict(vertex_attrs or {}).items()}
        self._earr = {}            # edge attributes as 2-d arrays, built on first use (edge_attribute_array)
        self.es = _EdgeSeq(self)
        self.vs = _VertexSeq(self)

    def edge_attribute_array(self, name):
        """The edge attribute as ONE [E, width] array in edge order -- what np.asarray(es.get_attribute_values(name)) gives,
        without the detour through 5 000 Python objects per batch (GraphConvInfo.set_batch_device concatenates these)."""
        a = self._earr.get(name)
        if a is None:
            vals = self._eattrs[name]
            a = np.asarray(vals) if len(vals) else np.zeros((0, 0), dtype=np.float32)
            self._earr[name] = a
        return a

    def get_edgelist(self):
        return [tuple(e) for e in self._edges.tolist()]

    def vcount(self):
        return self._n

    def ecount(self):
        return len(self._edges)

    def indegree(self, vs=None, loops=True):
        return np.bincount(self._edges[:, 1], minlength=self._n).tolist()

    def permute_vertices(self, perm):
        """Vertex k of this graph becomes vertex perm[k] of the result (igraph semantics); edges keep their order."""
        perm = np.asarray(perm, dtype=np.int64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(self._n)
        vattrs = {k: [v[i] for i in inv] for k, v in self._vattrs.items()}
        return SuperpointGraph(self._n, perm[self._edges] if len(self._edges) else self._edges, True, self._eattrs, vattrs)

    def neighborhood(self, centers, order=1):
        """For every center the vertices within `order` steps, direction ignored (igraph mode='all'), itself first."""
        from scipy import sparse
        n, e = self._n, self._edges
        adj = sparse.csr_matrix((np.ones(2 * len(e), dtype=np.int8), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])),
                                shape=(n, n))
        out = []
        for c in centers:
            seen = np.zeros(n, dtype=bool)
            seen[c] = True
            order_list, frontier = [int(c)], np.array([c])
            for _ in range(int(order)):
                nb = np.unique(adj[frontier].indices)
                nb = nb[~seen[nb]]
                if nb.size == 0:
                    break
                seen[nb] = True
                order_list += nb.tolist()
                frontier = nb
            out.append(order_list)
        return out

    def subgraph(self, vertices):
        """Induced subgraph; kept vertices are renumbered in increasing order of their old ids (igraph semantics, see the class
        docstring; every call site passes increasing ids anyway)."""
        ids = np.sort(np.asarray(list(vertices), dtype=np.int64))
        new_id = np.full(self._n, -1, dtype=np.int64)
        new_id[ids] = np.arange(len(ids))
        e = self._edges
        keep = np.nonzero((new_id[e[:, 0]] >= 0) & (new_id[e[:, 1]] >= 0))[0] if len(e) else np.zeros(0, dtype=np.int64)
        eattrs = {k: [v[i] for i in keep] for k, v in self._eattrs.items()}
        vattrs = {k: [v[i] for i in ids] for k, v in self._vattrs.items()}
        return SuperpointGraph(len(ids), new_id[e[keep]] if len(keep) else np.zeros((0, 2), dtype=np.int64), True, eattrs, vattrs)


def cloud_edge_feats(edgeattrs):
    """reference learning/spg.py:173-175"""
    edgefeats = np.asarray(edgeattrs['f'])
    return torch.from_numpy(edgefeats), None


def _as_tensor(x):
    return x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x))


def eccpc_collate(batch, device_batch=False):
    """Collates a list of dataset samples into a single batch (reference learning/spg.py:178-193).  Clouds produced by
    the device loader are CUDA tensors already and are concatenated on the device.
    device_batch: build the batched graph on the GPU (GraphConvInfo.set_batch_device: ordering by target, edge-feature
    reordering and the CSR / reverse CSR as kernels; the host only concatenates) -- for collation in the main process
    (`--loader_device 1`); the default is the reference's host construction (safe inside DataLoader workers)."""
    targets, graphs, clouds_meta, clouds_flag, clouds, clouds_global = list(zip(*batch))
    targets = torch.cat([_as_tensor(t) for t in targets if t is not None], 0).long()
    graphs = [graph for graph in graphs if graph is not None]
    has_clouds = len(clouds_meta[0]) > 0
    if has_clouds:
        clouds = torch.cat([_as_tensor(f) for f in clouds if f is not None], 0)
        clouds_global = torch.cat([_as_tensor(f) for f in clouds_global if f is not None], 0)
        clouds_flag = torch.cat([_as_tensor(f) for f in clouds_flag if f is not None], 0)
        clouds_meta = [item for sublist in clouds_meta if sublist is not None for item in sublist]
    if device_batch:
        # Everything small of the batch goes to the device in ONE staging copy together with the edge list and the edge features
        # (GraphConvInfo.set_batch_device(extras=...)): CloudEmbedder's two index vectors, the label vectors the trainer uploads
        # (learning/main.py:202-205: majority label / per-class point counts) and -- from a host-side loader -- the diameters.
        from .pointnet import attach_staged_flags, flag_index_vectors
        extras = [None, None, targets[:, 0].contiguous(), targets[:, 2:].contiguous(), None]
        if has_clouds:
            extras[0], extras[1] = flag_index_vectors(clouds_flag)
            if not clouds_global.is_cuda and not clouds_global.is_pinned():
                extras[4] = clouds_global
        gi = ecc.GraphConvInfo()
        gi.set_batch_device(graphs, cloud_edge_feats, extras=extras)
        GIs = [gi]
        dev_x = gi.extras_dev
        if has_clouds:
            attach_staged_flags(clouds_flag, dev_x[0], dev_x[1])
            if dev_x[4] is not None:
                clouds_global = dev_x[4]
        targets._spg_labels_dev = (dev_x[2], dev_x[3])      # (label_mode, label_vec) on the device: learning/main.py uses them if present
    else:
        GIs = [ecc.GraphConvInfo(graphs, cloud_edge_feats)]
    return targets, GIs, (clouds_meta, clouds_flag, clouds, clouds_global)


def sample_from_scene(scene, name='synthetic'):
    """A loader sample (reference learning/spg.py:166) from a synthetic scene of superpoint_graph_amd.synth."""
    G = SuperpointGraph(scene['n_sp'], scene['edges'], True, {'f': list(scene['edge_feats'])})
    meta = ['{}.{:d}'.format(name, i) for i in range(scene['n_sp'])]
    return scene['targets'], G, meta, scene['flag'], scene['clouds'], scene['diam']


# --------------------------------------------------------------------------------------------------------------------
# device-side load_superpoint / augment_cloud
# --------------------------------------------------------------------------------------------------------------------
_RAW_COLUMNS = {'xyz': (0, 1, 2), 'rgb': (3, 4, 5), 'e': (6,), 'lpsv': (7, 8, 9, 10), 'XYZ': (11, 12, 13)}


def pc_attribs_columns(pc_attribs):
    """Raw column of each output feature, in the reference's order (substring tests, learning/spg.py:224-232)."""
    if 'd' in pc_attribs:
        raise NotImplementedError("pc_attribs 'd' does not work in the reference either (1-D column, spg.py:231)")
    cols = []
    for key in ('xyz', 'rgb', 'e', 'lpsv', 'XYZ'):
        if key in pc_attribs:
            cols += _RAW_COLUMNS[key]
    return cols


_device_rng_step = [0]


def load_superpoints_device(args, points, offsets, ids, train, test_seed_offset=0, counts=None, rng=None):
    """All superpoints of a scene in ONE kernel launch: what `loader` does with one `load_superpoint` call per
    superpoint (reference learning/spg.py:150-167, 198-236) plus `augment_cloud` (:239-258) when `train`.

    points: device f32 [Ntot, ncols] raw rows of every superpoint back to back (the content of parsed/<scene>.h5),
    offsets: i64 [S+1] (host or device), ids: the superpoint ids (seed of the evaluation stream, :205).
    rng='host' (default; `args.loader_rng`): the random streams stay on the host and are consumed in the reference's
    order -- per superpoint: resampling (`rs.choice`, numpy), then the augmentation matrix (python `random`), then the
    jitter (numpy `randn`) -- so a seeded run yields the reference's clouds.
    rng='device': the streams come from a counter-based generator on the GPU (ops.loader_random), keyed by
    (args.seed [+ test_seed_offset in evaluation], superpoint id, call counter in training / 0 in evaluation): no host
    loop over the superpoints and no host->device copy of indices / noise; same distributions, different numbers.
    -> (clouds_flag i64[S] host, clouds f32[Nv, F, npts] device, clouds_global f32[Nv] device)
    """
    import math
    import random
    from .. import ops
    if not points.is_cuda:
        raise RuntimeError('superpoint_graph_amd.load_superpoints_device has no CPU path')
    off_h = offsets.cpu().numpy() if torch.is_tensor(offsets) else np.asarray(offsets, dtype=np.int64)
    if counts is None:                        # offsets [S+1]: superpoints back to back
        counts = np.diff(off_h)
    else:                                     # offsets [S] = first row of each (arbitrary subset / order of a resident scene)
        counts = np.asarray(counts, dtype=np.int64)
        off_h = np.concatenate([off_h, off_h[-1:] + counts[-1:]]) if len(off_h) else np.zeros(1, dtype=np.int64)
    S, npts = len(off_h) - 1, int(args.ptn_npts)
    cols = pc_attribs_columns(args.pc_attribs) if args.pc_attribs != '' else list(range(points.shape[1]))
    F = len(cols)
    flag = np.where(counts < args.ptn_minpts, -1, 0).astype(np.int64)           # :203
    slot = np.full(S, -1, dtype=np.int32)
    slot[flag == 0] = np.arange(int((flag == 0).sum()), dtype=np.int32)
    nv = int((flag == 0).sum())
    augment = bool(train)
    rng = rng or getattr(args, 'loader_rng', 'host')
    if rng == 'device':
        dev = points.device
        slot_d = ops.upload(torch.from_numpy(slot), dev)
        if train:
            _device_rng_step[0] += 1
        sidx_d, M_d, noise_d = ops.loader_random(
            ops.upload(torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)), dev),
            ops.upload(torch.from_numpy(np.asarray(ids, dtype=np.int64)), dev), slot_d, npts, F, nv,
            int(getattr(args, 'seed', 0)) + (0 if train else int(test_seed_offset)), _device_rng_step[0] if train else 0, augment,
            float(args.pc_augm_scale), args.pc_augm_rot == 1, float(args.pc_augm_mirror_prob),
            bool(getattr(args, 'pc_augm_jitter', 0)))
        clouds, diam = ops.load_superpoints(points, ops.upload(torch.from_numpy(off_h.astype(np.int64)), dev), slot_d, sidx_d, cols,
                                            bool(args.pc_xyznormalize), nv, M_d, noise_d)
        return torch.from_numpy(flag), clouds, diam
    if rng != 'host':
        raise ValueError(f"loader_rng must be 'host' or 'device', got {rng!r}")
    sidx = np.zeros((S, npts), dtype=np.int32)
    Ms = np.tile(np.eye(3), (S, 1, 1)) if augment else None
    jitter = augment and bool(getattr(args, 'pc_augm_jitter', 0))
    noise = np.zeros((nv, npts, F), dtype=np.float32) if jitter else None
    for s in range(S):
        n = int(counts[s])
        if flag[s] != 0:
            continue
        rs = np.random if train else np.random.RandomState(seed=int(ids[s]) + test_seed_offset)     # :205
        if n > npts:                                                                              # :207-214
            sidx[s] = rs.choice(n, npts)
        elif n < npts:
            sidx[s, :n] = np.arange(n)
            sidx[s, n:] = rs.choice(n, npts - n)
        else:
            sidx[s] = np.arange(n)
        if augment:                                                                               # :241-251
            M = np.eye(3)
            if args.pc_augm_scale > 1:
                M = np.dot(np.eye(3) * random.uniform(1 / args.pc_augm_scale, args.pc_augm_scale), M)
            if args.pc_augm_rot == 1:
                a = random.uniform(0, 2 * math.pi)
                c, sn = math.cos(a), math.sin(a)
                M = np.dot(np.array([[c, -sn, 0.0], [sn, c, 0.0], [0.0, 0.0, 1.0]]), M)           # z = upright
            if args.pc_augm_mirror_prob > 0:
                if random.random() < args.pc_augm_mirror_prob / 2:
                    M = np.dot(np.diag([-1.0, 1.0, 1.0]), M)
                if random.random() < args.pc_augm_mirror_prob / 2:
                    M = np.dot(np.diag([1.0, -1.0, 1.0]), M)
            Ms[s] = M
            if jitter:                                                                            # :255-257
                noise[slot[s]] = np.clip(0.01 * np.random.randn(npts, F), -0.05, 0.05).astype(np.float32)
    dev = points.device
    clouds, diam = ops.load_superpoints(
        points, ops.upload(torch.from_numpy(off_h.astype(np.int64)), dev), ops.upload(torch.from_numpy(slot), dev),
        ops.upload(torch.from_numpy(sidx), dev), cols, bool(args.pc_xyznormalize), nv,
        None if Ms is None else ops.upload(torch.from_numpy(Ms), dev), None if noise is None else ops.upload(torch.from_numpy(noise), dev))
    return torch.from_numpy(flag), clouds, diam


# --------------------------------------------------------------------------------------------------------------------
# superpoint-graph files -> graphs, edge features, loader  (reference learning/spg.py:23-171)
# --------------------------------------------------------------------------------------------------------------------
def spg_edge_features(edges, node_att, edge_att, args):
    """Assembles the superedge features from edge attributes and differences of node attributes (reference
    learning/spg.py:23-49): `delta_avg`, `delta_std` copied, `X/d` difference, `X/ld` log ratio, `X/r` ratio,
    `constant`; float32 result in the order of `args.edge_attribs`."""
    src, dst = edges[:, 0], edges[:, 1]
    cols = []
    for attrib in args.edge_attribs.split(','):
        a, _, opt = attrib.partition('/')
        opt = opt.lower()
        if a in ('delta_avg', 'delta_std'):
            cols.append(edge_att[a])
        elif a == 'constant':
            cols.append(np.ones((edges.shape[0], 1), dtype=np.float32))
        elif a in ('nlength', 'surface', 'volume', 'size', 'xyz'):
            v = node_att[a]
            if opt == 'd':
                v = v[src, :] - v[dst, :]
            elif opt == 'ld':
                v = np.log(v + 1e-10)
                v = v[src, :] - v[dst, :]
            elif opt == 'r':
                v = v[src, :] / (v[dst, :] + 1e-10)
            else:
                raise NotImplementedError(attrib)
            cols.append(v)
        else:
            raise NotImplementedError(attrib)
    return np.concatenate(cols, axis=1).astype(np.float32)


def spg_edge_features_device(edges, node_att, edge_att, args, scaler=None, device=None):
    """`spg_edge_features` (+ the `scaler01` transform when a fitted sklearn StandardScaler is given) on the GPU: one
    thread per (edge, feature); float32 node attributes are combined in float32 and the u64 point count in float64 with
    one final rounding, exactly like the numpy expressions of the reference (learning/spg.py:23-64).  Copies, differences
    and ratios are bit-identical to the host function; the logarithms are the device's logf / log."""
    from .. import ops
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device

    def up(a):
        a = np.asarray(a)
        a = a.astype(np.float64) if a.dtype not in (np.float32, np.float64) else a      # u64 counts promote like numpy does
        return torch.from_numpy(np.ascontiguousarray(a.reshape(a.shape[0], -1))).to(dev)
    cache, cols = {}, []
    for attrib in args.edge_attribs.split(','):
        a, _, opt = attrib.partition('/')
        opt = opt.lower()
        if a in ('delta_avg', 'delta_std'):
            t = cache.setdefault(a, up(edge_att[a]))
            cols += [('copy', t, c) for c in range(t.shape[1])]
        elif a == 'constant':
            cols.append(('const', None, 0))
        elif a in ('nlength', 'surface', 'volume', 'size', 'xyz'):
            if opt not in ('d', 'ld', 'r'):
                raise NotImplementedError(attrib)
            t = cache.setdefault(a, up(node_att[a]))
            cols += [(opt, t, c) for c in range(t.shape[1])]
        else:
            raise NotImplementedError(attrib)
    e = torch.from_numpy(np.ascontiguousarray(np.asarray(edges, dtype=np.int64))).to(dev)
    mean = scale = None
    if scaler is not None:
        mean = ops.upload(torch.from_numpy(np.asarray(scaler.mean_, dtype=np.float64)), dev)
        scale = ops.upload(torch.from_numpy(np.asarray(scaler.scale_, dtype=np.float64)), dev)
    return ops.edge_features(cols, e, mean, scale)


def scaler01(trainlist, testlist, transform_train=True, validlist=[]):
    """Standardises the edge features (column 3 of every `spg_reader` tuple) with the statistics of the TRAINING edges
    (reference learning/spg.py:51-64; sklearn's StandardScaler, in place)."""
    from sklearn import preprocessing
    scaler = preprocessing.StandardScaler().fit(np.concatenate([t[3] for t in trainlist], 0))
    for lst in ((trainlist if transform_train else []), testlist, validlist):
        for t in lst:
            scaler.transform(t[3], copy=False)
    return trainlist, testlist, validlist, scaler


def spg_from_arrays(args, f, name):
    """The superpoint-graph record of one scene from the arrays of its file (`f`: mapping name -> array with the datasets
    that partition/provider.py:558-600 writes) -> (node_gt, node_gt_size, edges, edge_feats, name); the body of the
    reference's spg_reader (learning/spg.py:68-103)."""
    sp_labels = np.asarray(f['sp_labels'])
    if sp_labels.size > 0:
        node_gt_size = sp_labels.astype(np.int64)               # col 0: unlabelled points, col 1+: points per class
        node_gt = np.argmax(node_gt_size[:, 1:], 1)[:, None]
        node_gt[node_gt_size[:, 1:].sum(1) == 0, :] = -100      # ignored by the loss
    else:
        n = np.asarray(f['sp_point_count']).shape[0]
        node_gt_size = np.concatenate([np.asarray(f['sp_point_count']).astype(np.int64), np.zeros((n, 8), dtype=np.int64)], 1)
        node_gt = np.zeros((n, 1), dtype=np.int64)
    node_att = dict(xyz=np.asarray(f['sp_centroids']), nlength=np.maximum(0, np.asarray(f['sp_length'])),
                    volume=np.maximum(0, np.asarray(f['sp_volume']) ** 2), surface=np.maximum(0, np.asarray(f['sp_surface']) ** 2),
                    size=np.asarray(f['sp_point_count']))
    edges = np.concatenate([np.asarray(f['source']), np.asarray(f['target'])], axis=1).astype(np.int64)
    edge_att = dict(delta_avg=np.asarray(f['se_delta_mean']), delta_std=np.asarray(f['se_delta_std']))
    if args.spg_superedge_cutoff > 0:
        keep = np.linalg.norm(edge_att['delta_avg'], axis=1) < args.spg_superedge_cutoff
        edges = edges[keep, :]
        edge_att = {k: v[keep, :] for k, v in edge_att.items()}
    return node_gt, node_gt_size, edges, spg_edge_features(edges, node_att, edge_att, args), name


def spg_reader(args, fname, incl_dir_in_name=False):
    """Loads a superpoint graph from its HDF5 file (reference learning/spg.py:66-103; needs h5py)."""
    import os
    import h5py
    name = os.path.basename(fname)[:-len('.h5')]
    if incl_dir_in_name:
        name = os.path.basename(os.path.dirname(fname)) + '/' + name
    with h5py.File(fname, 'r') as f:
        return spg_from_arrays(args, {k: f[k][:] for k in ('sp_labels', 'sp_centroids', 'sp_length', 'sp_volume', 'sp_surface',
                                                            'sp_point_count', 'source', 'target', 'se_delta_mean', 'se_delta_std')}, name)


def spg_to_graph(node_gt, node_gt_size, edges, edge_feats, fname, graph_cls=None):
    """`spg_to_igraph` (reference learning/spg.py:106-113): vertex attributes v (original id), t (targets row),
    s (point count).  `graph_cls` defaults to SuperpointGraph; igraph.Graph works as well."""
    targets = np.concatenate([node_gt, node_gt_size], axis=1)
    cls = graph_cls or SuperpointGraph
    G = cls(n=node_gt.shape[0], edges=edges.tolist(), directed=True, edge_attrs={'f': edge_feats},
            vertex_attrs={'v': list(range(node_gt.shape[0])), 't': targets, 's': node_gt_size.sum(1)})
    return G, fname


spg_to_igraph = spg_to_graph


def random_neighborhoods(G, num, order):
    """`num` random neighbourhoods of `order` hops, merged (reference learning/spg.py:115-122)."""
    import random
    centers = random.sample(range(G.vcount()), k=num)
    members = sorted({v for nb in G.neighborhood(centers, order) for v in nb})
    return G.subgraph(members)


def k_big_enough(G, minpts, k):
    """Induced graph on the leading vertices that contain at most k superpoints of >= minpts points (reference :124-128)."""
    big = np.cumsum(np.array(G.vs['s']) >= minpts)
    n = int(np.argwhere(big <= k)[-1][0]) + 1
    return G.subgraph(range(n))


class MemoryPointStore:
    """parsed/<scene>.h5 in memory: {scene name: {superpoint id: float array [n, ncols]}}."""

    def __init__(self, scenes):
        self._scenes = scenes

    def points(self, fname, sp_id):
        return self._scenes[fname][sp_id]

    def count(self, fname, sp_id):
        return self._scenes[fname][sp_id].shape[0]

    def ids(self, fname):
        return sorted(self._scenes[fname].keys())


class H5PointStore:
    """<db_path>/parsed/<scene>.h5 with one dataset per superpoint id (reference learning/spg.py:200-205; needs h5py)."""

    def __init__(self, db_path):
        self._root, self._open = db_path, {}

    def _file(self, fname):
        import h5py
        if fname not in self._open:
            self._open[fname] = h5py.File(self._root + '/parsed/' + fname + '.h5', 'r')
        return self._open[fname]

    def points(self, fname, sp_id):
        return self._file(fname)['{:d}'.format(sp_id)][:]

    def count(self, fname, sp_id):
        return self._file(fname)['{:d}'.format(sp_id)].shape[0]

    def ids(self, fname):
        return sorted(int(k) for k in self._file(fname).keys() if k.isdigit())


def load_superpoint(args, store, fname, sp_id, train, test_seed_offset):
    """One superpoint -> ([npts, F] float32 cloud or None, diameter) on the HOST, with the reference's random streams in
    the reference's order (learning/spg.py:198-258): the parity baseline of the device loader below and the path used when
    the model runs without a GPU-resident point store."""
    import math
    import random
    n = store.count(fname, sp_id)
    if n < args.ptn_minpts:
        return None, n
    P = np.asarray(store.points(fname, sp_id)).astype(np.float32)
    rs = np.random if train else np.random.RandomState(seed=sp_id + test_seed_offset)
    if n > args.ptn_npts:
        P = P[rs.choice(n, args.ptn_npts), ...]
    elif n < args.ptn_npts:
        P = np.concatenate([P, P[rs.choice(n, args.ptn_npts - n), ...]], 0)
    if args.pc_xyznormalize:
        diameter = np.max(np.max(P[:, :3], axis=0) - np.min(P[:, :3], axis=0))
        P[:, :3] = (P[:, :3] - np.mean(P[:, :3], axis=0, keepdims=True)) / (diameter + 1e-10)
    else:
        diameter = 0.0
        P[:, :3] = P[:, :3] - np.mean(P[:, :3], axis=0, keepdims=True)
    if args.pc_attribs != '':
        P = P[:, pc_attribs_columns(args.pc_attribs)]
    if train:
        M = np.eye(3)
        if args.pc_augm_scale > 1:
            M = np.dot(np.eye(3) * random.uniform(1 / args.pc_augm_scale, args.pc_augm_scale), M)
        if args.pc_augm_rot == 1:
            a = random.uniform(0, 2 * math.pi)
            c, sn = math.cos(a), math.sin(a)
            M = np.dot(np.array([[c, -sn, 0.0], [sn, c, 0.0], [0.0, 0.0, 1.0]]), M)
        if args.pc_augm_mirror_prob > 0:
            if random.random() < args.pc_augm_mirror_prob / 2:
                M = np.dot(np.diag([-1.0, 1.0, 1.0]), M)
            if random.random() < args.pc_augm_mirror_prob / 2:
                M = np.dot(np.diag([1.0, -1.0, 1.0]), M)
        P[:, :3] = np.dot(P[:, :3], M.T)
        if args.pc_augm_jitter:
            P = P + np.clip(0.01 * np.random.randn(*P.shape), -0.05, 0.05).astype(np.float32)
    return P, np.array([diameter], dtype=np.float32)


class DevicePointCache:
    """Scenes' parsed points resident in HBM (288 GB per GPU: the whole training set of S3DIS is a few GB): the first
    touch of a scene uploads its ragged point buffer once; afterwards a training step only sends indices."""

    def __init__(self, store, device):
        self._store, self._dev, self._scenes = store, device, {}

    def scene(self, fname):
        ent = self._scenes.get(fname)
        if ent is None:
            ids = self._store.ids(fname)
            arrs = [np.asarray(self._store.points(fname, i), dtype=np.float32) for i in ids]
            off = np.zeros(len(ids) + 1, dtype=np.int64)
            off[1:] = np.cumsum([a.shape[0] for a in arrs])
            pts = torch.from_numpy(np.concatenate(arrs, 0)).to(self._dev)
            ent = (pts, off, {i: k for k, i in enumerate(ids)})
            self._scenes[fname] = ent
        return ent


def loader(entry, train, args, db_path, test_seed_offset=0, store=None, device_cache=None):
    """Prepares a (possibly sub-sampled) superpoint graph and its superpoint clouds (reference learning/spg.py:130-171).
    `store`: point store (default: the HDF5 files under db_path); `device_cache`: DevicePointCache -> the clouds are built
    by ONE launch of the HIP loader kernel from the scene's resident ragged buffer and returned as CUDA tensors."""
    import random
    G, fname = entry
    if train:                                        # 1) neighbourhood sub-sampling of the (permuted) graph, :134-144
        if 0 < args.spg_augm_hardcutoff < G.vcount():
            perm = list(range(G.vcount()))
            random.shuffle(perm)
            G = G.permute_vertices(perm)
        if 0 < args.spg_augm_nneigh < G.vcount():
            G = random_neighborhoods(G, args.spg_augm_nneigh, args.spg_augm_order)
        if 0 < args.spg_augm_hardcutoff < G.vcount():
            G = k_big_enough(G, args.ptn_minpts, args.spg_augm_hardcutoff)
    if len(G.get_edgelist()) == 0:                   # graphs without edges are dropped by the collate, :169-171
        return None, None, None, None, None, None
    vids = [G.vs[s]['v'] for s in range(G.vcount())]
    meta = ['{}.{:d}'.format(fname, v) for v in vids]
    targets = np.array(G.vs['t'])
    if device_cache is not None:                     # 2) all clouds of the graph in one kernel launch
        pts, off, index = device_cache.scene(fname)
        rows = np.array([index[v] for v in vids], dtype=np.int64)
        flag, clouds, diam = load_superpoints_device(args, pts, off[rows], vids, train, test_seed_offset,
                                                     counts=off[rows + 1] - off[rows])
        return targets, G, meta, flag, clouds, diam
    store = store if store is not None else H5PointStore(db_path)
    flag, clouds, diams = [], [], []
    for v in vids:
        cloud, diam = load_superpoint(args, store, fname, v, train, test_seed_offset)
        if cloud is None:
            flag.append(-1)
        else:
            flag.append(0)
            clouds.append(cloud.T)
            diams.append(diam)
    flag = np.array(flag)
    clouds = np.stack(clouds) if clouds else clouds
    diams = np.concatenate(diams) if diams else diams
    return targets, G, meta, flag, clouds, diams
