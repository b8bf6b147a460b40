"""GraphNetwork: model-config mini-DSL -> module sequence (reference learning/graphnet.py:17-98), with the
RNN-ECC layers executed by the HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.init as init

from . import ecc  # noqa: F401  (kept for API parity: `learning.graphnet.ecc`)
from .modules import GRUCellEx, HipLinear, LSTMCellEx, RNNGraphConvModule


def create_fnet(widths, orthoinit, llbias, bnidx=-1):
    """Filter-generating network, a multi-layer perceptron (reference learning/graphnet.py:17-34; identical
    construction order, hence identical initialisation and state_dict keys)."""
    fnet_modules = []
    for k in range(len(widths) - 2):
        fnet_modules.append(nn.Linear(widths[k], widths[k + 1]))
        if orthoinit:
            init.orthogonal_(fnet_modules[-1].weight, gain=init.calculate_gain('relu'))
        if bnidx == k:
            fnet_modules.append(nn.BatchNorm1d(widths[k + 1]))
        fnet_modules.append(nn.ReLU(True))
    fnet_modules.append(nn.Linear(widths[-2], widths[-1], bias=llbias))
    if orthoinit:
        init.orthogonal_(fnet_modules[-1].weight)
    if bnidx == len(widths) - 1:
        fnet_modules.append(nn.BatchNorm1d(fnet_modules[-1].weight.size(0)))
    return nn.Sequential(*fnet_modules)


class GraphNetwork(nn.Module):
    """Constructed from the `config` string of comma-delimited layer tokens (reference
    learning/graphnet.py:37-98).  Supported tokens: f_K, b[_x], r, d_p, gru_R[_vv][_layernorm][_ingate][_catall]."""

    def __init__(self, config, nfeat, fnet_widths, fnet_orthoinit=True, fnet_llbias=True, fnet_bnidx=-1,
                 edge_mem_limit=1e20, use_pyg=True, cuda=True):
        super(GraphNetwork, self).__init__()
        self.gconvs = []
        for d, conf in enumerate(config.split(',')):
            conf = conf.strip().split('_')
            if conf[0] == 'f':
                self.add_module(str(d), HipLinear(nfeat, int(conf[1])))      # nn.Linear with HIP forward / backward
                nfeat = int(conf[1])
            elif conf[0] == 'b':
                self.add_module(str(d), nn.BatchNorm1d(nfeat, eps=1e-5, affine=len(conf) == 1))
            elif conf[0] == 'r':
                self.add_module(str(d), nn.ReLU(True))
            elif conf[0] == 'd':
                self.add_module(str(d), nn.Dropout(p=float(conf[1]), inplace=False))
            elif conf[0] == 'crf':
                raise NotImplementedError('crf_R (ECC-CRF) is out of scope: the reference implementation itself does '
                                          'not run on torch >= 1.5')
            elif conf[0] == 'gru' or conf[0] == 'lstm':
                nrepeats = int(conf[1])
                vv = bool(int(conf[2])) if len(conf) > 2 else True
                layernorm = bool(int(conf[3])) if len(conf) > 3 else True
                ingate = bool(int(conf[4])) if len(conf) > 4 else True
                cat_all = bool(int(conf[5])) if len(conf) > 5 else True
                fnet = create_fnet(fnet_widths + [nfeat ** 2 if not vv else nfeat], fnet_orthoinit, fnet_llbias, fnet_bnidx)
                if conf[0] == 'gru':
                    cell = GRUCellEx(nfeat, nfeat, bias=True, layernorm=layernorm, ingate=ingate)
                else:
                    cell = LSTMCellEx(nfeat, nfeat, bias=True, layernorm=layernorm, ingate=ingate)
                gconv = RNNGraphConvModule(cell, fnet, nfeat, vv=vv, nrepeats=nrepeats, cat_all=cat_all,
                                           edge_mem_limit=edge_mem_limit, use_pyg=use_pyg, cuda=cuda)
                self.add_module(str(d), gconv)
                self.gconvs.append(gconv)
                if cat_all:
                    nfeat *= nrepeats + 1
            elif len(conf[0]) > 0:
                raise NotImplementedError('Unknown module: ' + conf[0])

    def set_info(self, gc_infos, cuda):
        """Provides the convolution modules with the graph structure of the current batch."""
        gc_infos = gc_infos if isinstance(gc_infos, (list, tuple)) else [gc_infos]
        for i, gc in enumerate(self.gconvs):
            if cuda:
                gc_infos[i].cuda()
            gc.set_info(gc_infos[i])

    def forward(self, input):
        for module in self._modules.values():
            input = module(input)
        return input
