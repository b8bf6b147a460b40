"""GraphConvInfo: host-side batching of graphs into flat index buffers, same contract as the reference
(learning/ecc/GraphConvInfo.py:16-86): edges sorted by target, `idxn` = source node per edge, `degs` =
in-degree per node, edge features in the same order.  Integer outputs are bit-exact with the reference.
`.cuda()` additionally builds the device-side CSR / reverse-CSR used by the HIP kernels."""
from collections import defaultdict

import numpy as np
import torch


def _concat_edge_attribute(per_graph):
    """Per-graph edge-attribute arrays -> one array in batch order.  A graph WITHOUT edges (an isolated superpoint, a tiny
    scene) contributes an empty array whose trailing shape is unknown ((0, 0) from SuperpointGraph, (0,) from igraph's empty
    value list): it is skipped instead of being concatenated -- the reference and `set_batch` extend a Python list, for
    which an empty graph is harmless (GraphConvInfo.py:55-57).  All graphs empty: (0, 0)."""
    full = [np.asarray(a) for a in per_graph if np.asarray(a).shape[0] > 0]
    if not full:
        return np.zeros((0, 0), dtype=np.float32)
    return np.concatenate(full) if len(full) > 1 else full[0]


class GraphConvInfo(object):
    def __init__(self, *args, **kwargs):
        self._idxn = None
        self._idxe = None
        self._degrees = None
        self._degrees_gpu = None
        self._edgefeats = None
        self._edge_indexes = None
        self._parts = None          # node offsets of the batch's graphs (scenes), when the batch was built from graphs
        self._graph = None          # superpoint_graph_amd.ops.DeviceGraph after .cuda()
        if len(args) > 0 or len(kwargs) > 0:
            self.set_batch(*args, **kwargs)

    def set_batch(self, graphs, edge_feat_func):
        """graphs: igraph-like objects (get_edgelist(), es[...], es.attributes(), indegree(), vcount())."""
        graphs = graphs if isinstance(graphs, (list, tuple)) else [graphs]
        p = 0
        parts = [0]
        idxn, degrees, edge_indexes = [], [], []
        edgeattrs = defaultdict(list)
        for G in graphs:
            E = np.array(G.get_edgelist()).reshape(-1, 2)
            idx = E[:, 1].argsort()                      # sort by target (numpy default kind, as the reference)
            idxn.append(p + E[idx, 0])
            edgeseq = G.es[idx.tolist()]
            for a in G.es.attributes():
                edgeattrs[a] += edgeseq.get_attribute_values(a)
            degrees += G.indegree(G.vs, loops=True)
            edge_indexes.append(np.asarray(p + E[idx]))
            p += G.vcount()
            parts.append(p)
        self._parts = parts         # node offsets of the batch's graphs: [parts[k], parts[k+1]) is closed under edges
        self._edgefeats, self._idxe = edge_feat_func(edgeattrs)
        self._idxn = torch.LongTensor(np.concatenate(idxn))
        if self._idxe is not None:
            assert self._idxe.numel() == self._idxn.numel()
        self._degrees = torch.LongTensor(degrees)
        self._degrees_gpu = None
        self._edge_indexes = torch.LongTensor(np.concatenate(edge_indexes).T)
        self._graph = None

    def set_batch_device(self, graphs, edge_feat_func, device=None, extras=None):
        """`set_batch` with the ordering work on the GPU (spg_set_batch): the host only concatenates the edge lists and
        edge attributes in their original order, counts the in-degrees (the contract keeps a host copy of `degs`;
        utils.get_edge_shards reads it) and checks the endpoints -- no device-to-host copy, no synchronisation.  Buffers
        come out device-resident; the order inside a target segment is the STABLE one (the reference's numpy argsort
        leaves ties unspecified), everything else is identical to `set_batch`.
        extras: further small HOST tensors of the batch (CloudEmbedder's index vectors, labels, diameters ...; None entries allowed):
        they travel in the SAME staging copy as the edge list and the edge features (ops.upload_packed: one host-to-device copy per
        batch instead of one per vector); their device versions are left in `self.extras_dev`, in the order given."""
        from ... import ops
        graphs = graphs if isinstance(graphs, (list, tuple)) else [graphs]
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        p, edges = 0, []
        parts = [0]
        edgeattrs = defaultdict(list)
        for G in graphs:
            E = getattr(G, '_edges', None)
            E = np.asarray(G.get_edgelist()).reshape(-1, 2) if E is None else np.asarray(E).reshape(-1, 2)
            edges.append(E.astype(np.int64) + p)
            fast = getattr(G, 'edge_attribute_array', None)
            for a in G.es.attributes():
                # per-graph [E_g, width] arrays when the graph offers them (SuperpointGraph), else igraph's value lists
                edgeattrs[a].append(fast(a) if fast is not None else np.asarray(G.es.get_attribute_values(a)))
            p += G.vcount()
            parts.append(p)
        self._parts = parts
        edgeattrs = {a: _concat_edge_attribute(v) for a, v in edgeattrs.items()}
        edges_h = np.concatenate(edges) if edges else np.zeros((0, 2), dtype=np.int64)
        if edges_h.size and (int(edges_h.min()) < 0 or int(edges_h.max()) >= p):
            raise IndexError('GraphConvInfo.set_batch_device: an edge endpoint is outside [0, number of nodes)')
        self._degrees = torch.from_numpy(np.bincount(edges_h[:, 1], minlength=p).astype(np.int64))
        feats, self._idxe = edge_feat_func(edgeattrs)
        if self._idxe is not None:
            raise NotImplementedError('filter sharing (idxe) is not supported by set_batch_device')
        feats = feats.float()
        self._edge_indexes = None                                   # built on demand (get_pyg_buffers): only the pyg path reads it
        # a batch of a few scenes: ONE launch does the ordering by target, the edge-feature reordering and the CSR / reverse CSR,
        # fed from the host arrays through the staging ring (ops.batch_graph_build)
        extras = list(extras) if extras is not None else []
        self.extras_dev = [None] * len(extras)
        edges_t = torch.from_numpy(edges_h)
        if feats.dim() == 2 and not feats.is_cuda and ops.batch_graph_fits(p, int(edges_h.shape[0]), int(feats.shape[1])):
            # ONE staging copy for everything small of this batch, then the single-launch builder on the device copies
            packed = ops.upload_packed([edges_t, feats] + extras, dev)
            self.extras_dev = packed[2:]
            built = ops.batch_graph_build(packed[0], packed[1], p, dev)
            self._idxn, self._degrees_gpu, self._edgefeats, self._graph, _err = built
            return
        if extras:
            self.extras_dev = ops.upload_packed(extras, dev)
        built = ops.batch_graph_build(edges_t, feats if feats.dim() == 2 else None, p, dev) if feats.dim() == 2 else None
        if built is not None:
            self._idxn, self._degrees_gpu, self._edgefeats, self._graph, _err = built
            return
        # larger batches: the multi-launch path.  Uploads go through the staging ring: a copy from pageable memory blocks the host
        # until the stream reaches it, i.e. until the PREVIOUS step has drained -- host and GPU would take turns
        edges_d = ops.upload(torch.from_numpy(edges_h), dev)
        idxn, degs_gpu, perm, _err = ops.set_batch(edges_d, p)      # the device flag duplicates the host check above
        self._idxn, self._degrees_gpu = idxn, degs_gpu
        feats_d = ops.upload(feats, dev)
        self._edgefeats = ops.gather_rows(feats_d, perm) if idxn.numel() else feats_d
        self._graph = ops.DeviceGraph(self._idxn, self._degrees_gpu)

    @classmethod
    def from_buffers(cls, idxn, degs, edgefeats, idxe=None, edge_indexes=None, parts=None):
        """Build directly from already-batched buffers (synthetic scenes, tests).  parts: node offsets [0, n_0, n_0 + n_1, ...] of
        the batch's graphs, if known (lets the one-launch GRU recurrence serve batches above 2048 nodes scene by scene)."""
        gi = cls()
        gi._idxn, gi._degrees, gi._edgefeats, gi._idxe, gi._edge_indexes = idxn, degs, edgefeats, idxe, edge_indexes
        gi._parts = None if parts is None else [int(v) for v in parts]
        return gi

    def _validate(self):
        """Host-side check of the index contract before the device CSR is built from it (the reference's index_select
        would raise a device assert; the HIP graph build would write out of bounds): sum(degs) == E, 0 <= idxn < N."""
        idxn, degs = self._idxn, self._degrees
        if idxn.is_cuda or degs.is_cuda:
            return            # already-resident buffers (from_buffers with device tensors): checked by the caller
        N, E = int(degs.numel()), int(idxn.numel())
        if int(degs.sum()) != E or (N and int(degs.min()) < 0):
            raise ValueError(f'GraphConvInfo: sum(degs) = {int(degs.sum())} does not match the number of edges {E}')
        if E and (int(idxn.min()) < 0 or int(idxn.max()) >= N):
            raise IndexError(f'GraphConvInfo: idxn entries must be in [0, {N}), got [{int(idxn.min())}, {int(idxn.max())}]')
        if self._edgefeats is not None and self._idxe is None and self._edgefeats.shape[0] != E:
            raise ValueError(f'GraphConvInfo: {self._edgefeats.shape[0]} edge-feature rows for {E} edges')
        if self._parts is not None:
            # the scene boundaries are a promise to the one-launch recurrence ("every part is closed under edges": a wave only ever
            # waits for nodes of its own round) -- a wrong hint would show up as spin time-outs / wrong outputs much later
            parts = torch.as_tensor(self._parts, dtype=torch.int64)
            if parts.numel() < 2 or int(parts[0]) != 0 or int(parts[-1]) != N or bool((parts[1:] < parts[:-1]).any()):
                raise ValueError(f'GraphConvInfo: parts must be non-decreasing node offsets from 0 to {N}, got {self._parts}')
            if E:
                tgt = torch.repeat_interleave(torch.arange(N, dtype=torch.int64), degs.to(torch.int64))
                inner = parts[1:-1].contiguous()
                if bool((torch.bucketize(idxn.to(torch.int64), inner, right=True) != torch.bucketize(tgt, inner, right=True)).any()):
                    raise ValueError('GraphConvInfo: an edge crosses a part boundary (parts must be unions of connected components)')

    def cuda(self):
        from ... import ops
        if self._graph is not None and self._idxn.is_cuda:
            return            # built by set_batch_device: already resident
        self._validate()
        # reference GraphConvInfo.py:71-79; through the staging ring: pageable `.cuda()` copies stall the host until the stream has drained
        self._idxn = ops.upload(self._idxn)
        if self._idxe is not None:
            self._idxe = ops.upload(self._idxe)
        self._degrees_gpu = ops.upload(self._degrees)
        self._edgefeats = ops.upload(self._edgefeats)
        if self._edge_indexes is not None:
            self._edge_indexes = ops.upload(self._edge_indexes)
        self._graph = ops.DeviceGraph(self._idxn, self._degrees_gpu)

    def device_graph(self):
        if self._graph is None:
            raise RuntimeError('GraphConvInfo.cuda() must be called before the HIP graph convolution '
                               '(GraphNetwork.set_info(gc_infos, cuda=True) does it)')
        return self._graph

    def get_buffers(self):
        return self._idxn, self._idxe, self._degrees, self._degrees_gpu, self._edgefeats

    def get_pyg_buffers(self):
        if self._edge_indexes is None and self._idxn is not None and self._idxn.is_cuda:      # batch built by set_batch_device
            tgt = torch.repeat_interleave(torch.arange(self._degrees.numel()), self._degrees).to(self._idxn.device)
            self._edge_indexes = torch.stack([self._idxn, tgt], 0)
        return self._edge_indexes
