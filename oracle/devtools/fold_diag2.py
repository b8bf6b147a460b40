"""Diagnostic: are the batch statistics of the fold path (fixed-point slots) equal to the float64 statistics of the raw layer
outputs the SAME run produced?  Stand-alone STN (two conv layers) on n clouds of 20 points.  GPU only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import validate_against_reference as V
from superpoint_graph_amd import _lib, ops
from superpoint_graph_amd.learning import pointnet
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65535
model = V.make_local_model(pointnet)
stn = model.stn.cuda().train()
clouds, cg, w = V.local_inputs(n)
x = clouds[:, :2, :].contiguous().cuda()
for mode in (0, 1):
    L.spg_tune(10, mode)
    cfg = stn._cfg(x.shape[2])
    emb, st = ops.pointnet_forward(cfg, x, None, stn._groups_tensors(), True, 1)
    torch.cuda.synchronize()
    B, Pn = x.shape[0], x.shape[2]
    nl = L.spg_pointnet_num_layers(ctypes.byref(cfg))
    widths = list(stn._nf_conv) + list(stn._nf_fc)
    for li, c in enumerate(widths):
        rows = B * Pn if li < len(stn._nf_conv) else B
        def buf(what, m):
            off = L.spg_pointnet_debug_offset(ctypes.byref(cfg), B, 1, li, what)
            return st.ws[off:off + 4 * m].view(torch.float32)
        y = buf(0, rows * c).view(rows, c).double()
        mean, rstd = buf(3, c).double(), buf(4, c).double()
        tm, tv = y.mean(0), y.var(0, unbiased=False)
        print(f'fold={1 - mode} layer {li} [{rows} x {c}]: mean err {float((mean - tm).abs().max() / tm.abs().max()):.2e}  '
              f'rstd err {float((rstd - 1 / torch.sqrt(tv + 1e-5)).abs().max() / (1 / torch.sqrt(tv + 1e-5)).abs().max()):.2e}')
L.spg_tune(10, 0)
