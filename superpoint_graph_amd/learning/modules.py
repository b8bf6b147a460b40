"""RNN-ECC module and GRU / LSTM cells with the reference's module API (learning/modules.py:128-316)."""
import torch
import torch.nn as nn

from .. import ops


class GRUCellEx(nn.GRUCell):
    """GRU cell extended with row normalisation of the gate pre-activations and an input gate (reference
    learning/modules.py:205-259; same parameters: weight_ih, weight_hh, bias_ih, bias_hh, ig.*).
    forward/backward run in one fused HIP kernel each (32 channels)."""

    def __init__(self, input_size, hidden_size, bias=True, layernorm=True, ingate=True):
        super(GRUCellEx, self).__init__(input_size, hidden_size, bias)
        self._layernorm = layernorm
        self._ingate = ingate
        if layernorm:
            self.add_module('ini', nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False))
            self.add_module('inh', nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False))
        if ingate:
            self.add_module('ig', nn.Linear(hidden_size, input_size, bias=True))

    def param_tensors(self):
        ig = self._modules['ig'] if self._ingate else None
        return (self.weight_ih, self.weight_hh, self.bias_ih, self.bias_hh,
                None if ig is None else ig.weight, None if ig is None else ig.bias)

    def forward(self, input, hidden):
        if not input.is_cuda:
            raise RuntimeError('superpoint_graph_amd.GRUCellEx has no CPU path')
        if self.input_size != 32 or self.hidden_size != 32 or self.bias_ih is None:
            raise NotImplementedError('the HIP GRU cell is specialised for 32 channels with bias')
        params = [p for p in self.param_tensors() if p is not None]
        return _GRUCellFunction.apply(self, input.contiguous(), hidden.contiguous(), *params)

    def __repr__(self):
        s = super(GRUCellEx, self).__repr__() + '('
        if self._ingate:
            s += 'ingate'
        if self._layernorm:
            s += ' layernorm'
        return s + ')'


class _GRUCellFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cell, input, hidden, *params):
        ctx.cell = cell
        ctx.save_for_backward(input, hidden)
        return ops.gru_cell_fwd(input, hidden, cell.param_tensors(), cell._layernorm, cell._ingate)

    @staticmethod
    def backward(ctx, grad_out):
        input, hidden = ctx.saved_tensors
        cell = ctx.cell
        gi, gh, grads = ops.gru_cell_bwd(input, hidden, grad_out, cell.param_tensors(), cell._layernorm, cell._ingate)
        return (None, gi, gh) + tuple(g for g in grads if g is not None)


class LSTMCellEx(nn.LSTMCell):
    """LSTM cell extended with row normalisation of the gate pre-activations and an input gate (reference
    learning/modules.py:262-316; same parameters).  forward(input, (h, c)) -> (hy, cy), one fused HIP kernel each way."""

    def __init__(self, input_size, hidden_size, bias=True, layernorm=True, ingate=True):
        super(LSTMCellEx, self).__init__(input_size, hidden_size, bias)
        self._layernorm = layernorm
        self._ingate = ingate
        if layernorm:
            self.add_module('ini', nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False))
            self.add_module('inh', nn.InstanceNorm1d(1, eps=1e-5, affine=False, track_running_stats=False))
        if ingate:
            self.add_module('ig', nn.Linear(hidden_size, input_size, bias=True))

    param_tensors = GRUCellEx.param_tensors

    def forward(self, input, hidden):
        if not input.is_cuda:
            raise RuntimeError('superpoint_graph_amd.LSTMCellEx has no CPU path')
        if self.input_size != 32 or self.hidden_size != 32 or self.bias_ih is None:
            raise NotImplementedError('the HIP LSTM cell is specialised for 32 channels with bias')
        params = [p for p in self.param_tensors() if p is not None]
        return _LSTMCellFunction.apply(self, input.contiguous(), hidden[0].contiguous(), hidden[1].contiguous(), *params)

    def __repr__(self):
        s = super(LSTMCellEx, self).__repr__() + '('
        if self._ingate:
            s += 'ingate'
        if self._layernorm:
            s += ' layernorm'
        return s + ')'


class _LSTMCellFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cell, input, h, c, *params):
        ctx.cell = cell
        ctx.save_for_backward(input, h, c)
        return ops.lstm_cell_fwd(input, h, c, cell.param_tensors(), cell._layernorm, cell._ingate)

    @staticmethod
    def backward(ctx, grad_hy, grad_cy):
        input, h, c = ctx.saved_tensors
        cell = ctx.cell
        gi, gh, gc, grads = ops.lstm_cell_bwd(input, h, c, grad_hy, grad_cy, cell.param_tensors(), cell._layernorm,
                                              cell._ingate)
        return (None, gi, gh, gc) + tuple(g for g in grads if g is not None)


class _LinearFunction(torch.autograd.Function):
    """nn.Linear on the few-row MFMA kernels: y = x W^T + b; backward = data gradient, weight gradient, bias column sum."""

    @staticmethod
    def forward(ctx, module, x, weight, bias):
        ctx.module = module
        ctx.save_for_backward(x, weight)
        return ops.linear_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        m = ctx.module
        gy = gy.contiguous()
        need_dx = ctx.needs_input_grad[1]
        direct = getattr(m, '_spg_direct_grads', False) and weight.grad is not None and weight.grad.is_contiguous() and \
            (m.bias is None or (m.bias.grad is not None and m.bias.grad.is_contiguous()))
        if direct:          # FlatParameters: write into the arena views, nothing for autograd to accumulate
            from ..flat import mark_direct_write
            mark_direct_write(m)
            gx, _, _ = ops.linear_backward(gy, x, weight, need_dx, m.bias is not None, out_w=weight.grad,
                                           out_b=None if m.bias is None else m.bias.grad)
            return None, gx, None, None
        if getattr(m, '_spg_direct_grads', False):
            if any(p is not None and p.requires_grad and (p.grad is None or not p.grad.is_contiguous()) for p in (weight, m.bias)):
                raise RuntimeError('FlatParameters mode: a parameter has no contiguous .grad view into the gradient arena '
                                   '(optimizer.zero_grad(set_to_none=True)?); use FlatParameters.zero_grad()')
            from ..flat import prepare_autograd_fallback       # partly frozen layer: autograd accumulates what is returned
            prepare_autograd_fallback(m)
        gx, gw, gb = ops.linear_backward(gy, x, weight, need_dx, m.bias is not None)
        return None, gx, gw, gb


class HipLinear(nn.Linear):
    """`nn.Linear` (same parameters / state_dict keys / initialisation) whose forward and backward run on the HIP kernels
    for 2-d float32 CUDA inputs -- the classifier of the graph network (reference learning/graphnet.py:47-49).  torch's
    path costs three hipBLASLt launches with host-side argument uploads, a reduction and two accumulations per step."""

    def _kernel_shape_ok(self):
        w = self.weight
        return w.is_contiguous() and w.shape[1] % 4 == 0 and w.data_ptr() % 16 == 0

    def forward(self, input):
        if input.is_cuda and input.dim() == 2 and input.dtype == torch.float32 and self._kernel_shape_ok():
            return _LinearFunction.apply(self, input.contiguous(), self.weight, self.bias)
        if getattr(self, '_spg_direct_grads', False) and torch.is_grad_enabled():
            from ..flat import prepare_autograd_fallback       # torch's path accumulates: see FlatParameters(lazy_zero=True)
            prepare_autograd_fallback(self)
        return super(HipLinear, self).forward(input)


def _fnet_groups(fnet):
    """(Linear, BatchNorm-or-None) pairs of a create_fnet() Sequential, plus its bnidx."""
    groups, bnidx, mods = [], -1, list(fnet)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            if bn is not None:
                bnidx = len(groups)
            groups.append((m, bn))
        elif not isinstance(m, (nn.BatchNorm1d, nn.ReLU)):
            raise NotImplementedError(f'unsupported filter-network module {type(m).__name__}')
    return groups, bnidx


class _EccRnnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, hx, edgefeats, graph, training, *flat_params):
        cfg, groups = module._cfg_for(module._gci, graph.N)
        out, state = ops.eccrnn_forward(cfg, graph, hx.contiguous(), edgefeats, groups, training, 1)
        ctx.module, ctx.state, ctx.groups, ctx.nflat = module, state, groups, len(flat_params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nf = ctx.state.cfg.n_fnet
        from .pointnet import _direct_grad_targets
        d_f = _direct_grad_targets(ctx.module, ctx.groups[:nf], 4)
        d_c = _direct_grad_targets(ctx.module, ctx.groups[nf:], 6)
        if d_f is not None and d_c is not None:
            from ..flat import mark_direct_write
            mark_direct_write(ctx.module)
            grad_h0, _ = ops.eccrnn_backward(ctx.state, ctx.groups, grad_out, d_f + d_c)
            return (None, grad_h0, None, None, None) + (None,) * ctx.nflat
        if ctx.nflat == 1 and getattr(ctx.module, '_spg_direct_grads', False):
            raise RuntimeError('FlatParameters mode: a parameter has no contiguous .grad view into the gradient arena '
                               '(optimizer.zero_grad(set_to_none=True) or a frozen parameter?); use FlatParameters.zero_grad()')
        grad_h0, gg = ops.eccrnn_backward(ctx.state, ctx.groups, grad_out)
        flat = []
        for li, g in enumerate(gg):
            flat += list(g[:4]) if li < nf else list(g)
        flat = [t for t in flat if t is not None]
        return (None, grad_h0, None, None, None) + tuple(flat)


class RNNGraphConvModule(nn.Module):
    """Recurrent graph convolution: filter-generating network once, then nrepeats x {ECC, RNN cell}
    (reference learning/modules.py:128-183; same constructor signature, `_cell` / `_fnet` attribute names).
    With a GRUCellEx / LSTMCellEx cell and 32 channels the whole module is one C call (spg_eccrnn_forward)."""

    def __init__(self, cell, filter_net, nfeat=None, vv=True, gc_info=None, nrepeats=1, cat_all=False,
                 edge_mem_limit=1e20, use_pyg=True, cuda=True):
        super(RNNGraphConvModule, self).__init__()
        self._cell = cell
        self._isLSTM = 'LSTM' in type(cell).__name__
        self._fnet = filter_net
        self._nrepeats = nrepeats
        self._cat_all = cat_all
        self._edge_mem_limit = edge_mem_limit
        self._vv = vv
        self.set_info(gc_info)
        self.use_pyg = use_pyg
        if use_pyg:
            raise NotImplementedError('--use_pyg 1 (torch_geometric NNConv) is out of scope of the HIP path; use --use_pyg 0')

    def set_info(self, gc_info):
        self._gci = gc_info

    def _cfg_and_groups(self):
        cached = self.__dict__.get('_cg_cache')
        if cached is None:
            cached = self._build_cfg_and_modules()
            self.__dict__['_cg_cache'] = cached
        cfg, fg = cached
        groups = []
        for lin, b in fg:
            groups.append((lin.weight, lin.bias, None if b is None else b.weight, None if b is None else b.bias,
                           None if b is None else b.running_mean, None if b is None else b.running_var))
        groups.append(self._cell.param_tensors())
        return cfg, groups

    def _cfg_for(self, gci, n_nodes):
        """(cfg, groups) of this batch: the cached configuration, with the batch's scene boundaries attached when the graph is
        too large for one round of the one-launch GRU recurrence (include/spg_hip.h: spg_eccrnn_cfg.n_parts) -- whole scenes are
        then processed in rounds instead of one launch per iteration.  A copy per distinct partition: the configuration object of
        a forward travels with its saved state to the backward."""
        cfg, groups = self._cfg_and_groups()
        parts = getattr(gci, '_parts', None)
        if parts is None or n_nodes <= 2048 or len(parts) - 1 > ops._lib.SPG_MAX_PARTS or len(parts) < 3:
            return cfg, groups
        key = tuple(parts)
        cache = self.__dict__.setdefault('_cfg_parts_cache', {})
        c2 = cache.get(key)
        if c2 is None:
            c2 = type(cfg).from_buffer_copy(cfg)
            c2.n_parts = len(parts) - 1
            for i, v in enumerate(parts):
                c2.part_ptr[i] = int(v)
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cache[key] = c2
        return c2, groups

    def _build_cfg_and_modules(self):
        fg, bnidx = _fnet_groups(self._fnet)
        widths = [fg[0][0].in_features] + [lin.out_features for lin, _ in fg]
        nc = self._cell.hidden_size
        matrix = widths[-1] == nc * nc and widths[-1] != nc
        bn = next((b for _, b in fg if b is not None), None)
        if bn is not None:
            from .pointnet import _bn_momentum
            _bn_momentum(bn)
        cfg = ops.make_eccrnn_cfg(nc, self._nrepeats, matrix, self._cell._layernorm, self._cell._ingate, self._cat_all,
                                  widths, bnidx, fg[-1][0].bias is not None,
                                  1e-5 if bn is None else bn.eps, 0.1 if bn is None or bn.momentum is None else bn.momentum,
                                  cell='lstm' if self._isLSTM else 'gru')
        return cfg, fg

    def _flat_params(self):
        flat = []
        for lin, b in _fnet_groups(self._fnet)[0]:
            flat += [lin.weight, lin.bias] + ([b.weight, b.bias] if b is not None else [])
        flat += list(self._cell.param_tensors())
        return [p for p in flat if p is not None]

    def forward(self, hx):
        if not hx.is_cuda:
            raise RuntimeError('superpoint_graph_amd.RNNGraphConvModule has no CPU path')
        idxn, idxe, degs, degs_gpu, edgefeats = self._gci.get_buffers()
        if idxe is not None:
            raise NotImplementedError('filter sharing (idxe) is not supported by the fused RNN-ECC path')
        if self.training:
            for m in self._fnet:
                if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None:
                    m.num_batches_tracked += 1
        if getattr(self, '_spg_direct_grads', False) and torch.is_grad_enabled():
            from .pointnet import _grad_anchor
            extra = (_grad_anchor(hx.device),)      # FlatParameters mode: one differentiable anchor instead of all parameters
        else:
            extra = tuple(self._flat_params())
        return _EccRnnFunction.apply(self, hx, edgefeats.contiguous().float(), self._gci.device_graph(), self.training, *extra)
