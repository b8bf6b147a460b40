"""GraphNetwork: the model-config mini-language -> a sequence of modules, with the RNN-ECC layers and the dense layers
executed by the HIP kernels.  Interface, state_dict keys and parameter initialisation (the order in which modules are
constructed and hence the order in which torch's RNG is consumed) follow reference learning/graphnet.py:17-98."""
import torch  # noqa: F401
import torch.nn as nn
from torch.nn import init

from . import ecc  # noqa: F401  (API parity: `learning.graphnet.ecc`)
from .modules import GRUCellEx, HipLinear, LSTMCellEx, RNNGraphConvModule


def create_fnet(widths, orthoinit, llbias, bnidx=-1):
    """Filter-generating MLP over the superedge features (reference learning/graphnet.py:17-34): Linear(+BatchNorm at
    layer `bnidx`)+ReLU for every hidden width, then a Linear with optional bias; orthogonal initialisation (gain of
    ReLU for the hidden layers) when `orthoinit`."""
    n_hidden = len(widths) - 2
    layers = []
    for idx, (fan_in, fan_out) in enumerate(zip(widths[:n_hidden], widths[1:n_hidden + 1])):
        lin = nn.Linear(fan_in, fan_out)
        if orthoinit:
            init.orthogonal_(lin.weight, gain=init.calculate_gain('relu'))
        layers.append(lin)
        if idx == bnidx:
            layers.append(nn.BatchNorm1d(fan_out))
        layers.append(nn.ReLU(True))
    head = nn.Linear(widths[-2], widths[-1], bias=llbias)
    if orthoinit:
        init.orthogonal_(head.weight)
    layers.append(head)
    if bnidx == len(widths) - 1:
        layers.append(nn.BatchNorm1d(widths[-1]))
    return nn.Sequential(*layers)


def _flag(tokens, pos):
    """optional 0/1 field of a layer token; absent = on"""
    return bool(int(tokens[pos])) if pos < len(tokens) else True


class GraphNetwork(nn.Module):
    """Built from a comma-separated list of layer tokens (reference learning/graphnet.py:37-98):
    `f_K` linear to K, `b[_x]` BatchNorm (non-affine with a suffix), `r` ReLU, `d_p` dropout,
    `gru_R[_vv][_layernorm][_ingate][_catall]` / `lstm_R[...]` R iterations of ECC + recurrent cell."""

    _CELLS = {'gru': GRUCellEx, 'lstm': LSTMCellEx}

    def __init__(self, config, nfeat, fnet_widths, fnet_orthoinit=True, fnet_llbias=True, fnet_bnidx=-1,
                 edge_mem_limit=1e20, use_pyg=True, cuda=True):
        super().__init__()
        self.gconvs = []
        width = nfeat
        for position, token in enumerate(config.split(',')):
            fields = token.strip().split('_')
            kind = fields[0]
            if kind == '':
                continue
            if kind == 'f':
                layer, width = HipLinear(width, int(fields[1])), int(fields[1])      # nn.Linear, HIP forward / backward
            elif kind == 'b':
                layer = nn.BatchNorm1d(width, eps=1e-5, affine=(len(fields) == 1))
            elif kind == 'r':
                layer = nn.ReLU(True)
            elif kind == 'd':
                layer = nn.Dropout(p=float(fields[1]), inplace=False)
            elif kind in self._CELLS:
                layer, width = self._recurrent_conv(kind, fields, width, fnet_widths, fnet_orthoinit, fnet_llbias, fnet_bnidx,
                                                    edge_mem_limit, use_pyg, cuda)
                self.gconvs.append(layer)
            elif kind == 'crf':
                raise NotImplementedError('crf_R (ECC-CRF) is out of scope: the reference implementation itself does '
                                          'not run on torch >= 1.5')
            else:
                raise NotImplementedError('Unknown module: ' + kind)
            self.add_module(str(position), layer)

    def _recurrent_conv(self, kind, fields, width, fnet_widths, orthoinit, llbias, bnidx, edge_mem_limit, use_pyg, cuda):
        repeats = int(fields[1])
        vector_filters, layernorm, ingate, cat_all = (_flag(fields, k) for k in (2, 3, 4, 5))
        # the filter network first, then the cell: the order fixes the random initialisation
        fnet = create_fnet(fnet_widths + [width if vector_filters else width * width], orthoinit, llbias, bnidx)
        cell = self._CELLS[kind](width, width, bias=True, layernorm=layernorm, ingate=ingate)
        conv = RNNGraphConvModule(cell, fnet, width, vv=vector_filters, nrepeats=repeats, cat_all=cat_all,
                                  edge_mem_limit=edge_mem_limit, use_pyg=use_pyg, cuda=cuda)
        return conv, width * (repeats + 1) if cat_all else width

    def set_info(self, gc_infos, cuda):
        """Hands the graph structure of the current batch to the convolution layers (one info object per layer)."""
        infos = list(gc_infos) if isinstance(gc_infos, (list, tuple)) else [gc_infos]
        for conv, info in zip(self.gconvs, infos):
            if cuda:
                info.cuda()
            conv.set_info(info)

    def forward(self, input):
        x = input
        for layer in self.children():
            x = layer(x)
        return x
