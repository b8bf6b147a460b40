"""Batching contract of the reference's learning/spg.py for the hot path: `eccpc_collate` turns a list of
loader samples `(targets, graph, clouds_meta, clouds_flag, clouds, clouds_global)` (reference
learning/spg.py:130-171) into `(targets, [GraphConvInfo], (clouds_meta, clouds_flag, clouds,
clouds_global))` (reference learning/spg.py:178-193).  File readers / augmentation are out of scope;
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


class SuperpointGraph:
    """Directed graph with edge attributes: the subset of igraph.Graph used by
    learning/ecc/GraphConvInfo.py:48-58 (get_edgelist, es[...], es.attributes, indegree, vcount, vs)."""

    def __init__(self, n, edges, edge_attrs=None, vertex_attrs=None):
        self._n = int(n)
        self._edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        self._eattrs = dict(edge_attrs or {})
        self._vattrs = dict(vertex_attrs or {})
        self.es = _EdgeSeq(self)
        self.vs = list(range(self._n))

    def get_edgelist(self):
        return [tuple(e) for e in self._edges.tolist()]

    def vcount(self):
        return self._n

    def indegree(self, vs=None, loops=True):
        return np.bincount(self._edges[:, 1], minlength=self._n).tolist()


def cloud_edge_feats(edgeattrs):
    """reference learning/spg.py:173-175"""
    edgefeats = np.asarray(edgeattrs['f'])
    return torch.from_numpy(edgefeats), None


def eccpc_collate(batch):
    """Collates a list of dataset samples into a single batch (reference learning/spg.py:178-193)."""
    targets, graphs, clouds_meta, clouds_flag, clouds, clouds_global = list(zip(*batch))
    targets = torch.cat([torch.from_numpy(t) for t in targets if t is not None], 0).long()
    graphs = [graph for graph in graphs if graph is not None]
    GIs = [ecc.GraphConvInfo(graphs, cloud_edge_feats)]
    if len(clouds_meta[0]) > 0:
        clouds = torch.cat([torch.from_numpy(f) for f in clouds if f is not None], 0)
        clouds_global = torch.cat([torch.from_numpy(f) for f in clouds_global if f is not None], 0)
        clouds_flag = torch.cat([torch.from_numpy(f) for f in clouds_flag if f is not None], 0)
        clouds_meta = [item for sublist in clouds_meta if sublist is not None for item in sublist]
    return targets, GIs, (clouds_meta, clouds_flag, clouds, clouds_global)


def sample_from_scene(scene, name='synthetic'):
    """A loader sample (reference learning/spg.py:166) from a synthetic scene of superpoint_graph_amd.synth."""
    G = SuperpointGraph(scene['n_sp'], scene['edges'], {'f': list(scene['edge_feats'])})
    meta = ['{}.{:d}'.format(name, i) for i in range(scene['n_sp'])]
    return scene['targets'], G, meta, scene['flag'], scene['clouds'], scene['diam']
