"""GraphConvFunction: the edge-conditioned convolution operator with the reference's call signature
(learning/ecc/GraphConvModule.py:19-152), executed by libspg_hip (one fused launch instead of
index_select + bmm + conv_aggregate per shard; `edge_mem_limit` is accepted and ignored: results do not
depend on the sharding)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops

_graph_cache = {}


def _graph_for(idxn, degs_gpu, n_src):
    """Device CSR for (idxn, degs) buffers handed to GraphConvFunction.apply directly (GraphConvInfo.cuda()
    normally builds it once per batch); cached by buffer identity."""
    key = (idxn.data_ptr(), degs_gpu.data_ptr(), idxn.numel(), degs_gpu.numel(), n_src)
    g = _graph_cache.get(key)
    if g is None or g.idxn is not idxn:
        if len(_graph_cache) > 8:
            _graph_cache.clear()
        g = ops.DeviceGraph(idxn, degs_gpu, n_src)
        _graph_cache[key] = g
    return g


class GraphConvFunction(Function):
    """out[i] = mean_{edges e into i} x[idxn[e]] @ W[e]   (3-D weights)   or   x[idxn[e]] * w[e]   (2-D weights);
    rows with in-degree 0 are exactly 0."""

    @staticmethod
    def forward(ctx, input, weights, in_channels, out_channels, idxn, idxe, degs, degs_gpu, edge_mem_limit=1e20):
        if not input.is_cuda:
            raise RuntimeError('superpoint_graph_amd.GraphConvFunction has no CPU path; move tensors to the GPU')
        full = weights.dim() == 3
        assert full or (in_channels == out_channels and weights.size(1) == in_channels)
        if degs_gpu is None:
            degs_gpu = degs.to(input.device)
        graph = degs_gpu if isinstance(degs_gpu, ops.DeviceGraph) else _graph_for(idxn, degs_gpu, input.shape[0])
        input, weights = input.contiguous(), weights.contiguous()
        ctx.save_for_backward(input, weights)
        ctx._graph, ctx._idxe = graph, idxe
        return ops.ecc_aggregate_fwd(input, weights, graph, idxe, in_channels, out_channels)

    @staticmethod
    def backward(ctx, grad_output):
        input, weights = ctx.saved_tensors
        gx, gw = ops.ecc_aggregate_bwd(input, weights, grad_output, ctx._graph, ctx._idxe,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None, None, None, None, None, None


class GraphConvModule(nn.Module):
    """Reference learning/ecc/GraphConvModule.py:156-193: ECC with a filter-generating network."""

    def __init__(self, in_channels, out_channels, filter_net, gc_info=None, edge_mem_limit=1e20):
        super(GraphConvModule, self).__init__()
        self._in_channels = in_channels
        self._out_channels = out_channels
        self._fnet = filter_net
        self._edge_mem_limit = edge_mem_limit
        self.set_info(gc_info)

    def set_info(self, gc_info):
        self._gci = gc_info

    def forward(self, input):
        idxn, idxe, degs, degs_gpu, edgefeats = self._gci.get_buffers()
        weights = self._fnet(edgefeats)
        assert input.dim() == 2 and weights.dim() == 2
        if weights.size(1) == self._in_channels * self._out_channels:
            weights = weights.view(-1, self._in_channels, self._out_channels)
        return GraphConvFunction.apply(input, weights, self._in_channels, self._out_channels, idxn, idxe, degs,
                                       self._gci.device_graph(), self._edge_mem_limit)
