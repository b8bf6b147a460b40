"""Tensor-level wrappers over the C ABI (include/spg_hip.h): torch owns device memory and streams, the HIP
library does the arithmetic.  Every function here requires CUDA(ROCm) tensors and raises otherwise --
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import EccRnnCfg, PointNetCfg, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _req(t: torch.Tensor, dtype=None, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on the GPU: the superpoint_graph_amd kernels have no CPU path')
    if t.device.index != torch.cuda.current_device():
        # the kernels are launched on the CURRENT device's stream; a tensor of another GPU would be an invalid access
        raise RuntimeError(f'{name} lives on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}; '
                           'call torch.cuda.set_device() (or use `with torch.cuda.device_of(tensor):`) first')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    return t


def _ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


# --------------------------------------------------------------------------------------------------
# host -> device staging
# --------------------------------------------------------------------------------------------------
def upload(host: torch.Tensor, device=None) -> torch.Tensor:
    """Asynchronous host-to-device copy of a small per-batch tensor (edge lists, index vectors, edge features) through the
    library's page-locked staging ring (spg_upload): a copy from PAGEABLE memory blocks the host until the stream reaches it
    -- i.e. until the previous step has drained, so host and GPU would take turns -- and pinning per batch costs tens of
    milliseconds.  The call returns once the copy is enqueued on the current stream; `host` may be re-used at once."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if host.is_cuda:
        return host.to(device)
    if host.is_pinned():                      # page-locked already (DataLoader pin_memory, pre-pinned buffers): a true asynchronous copy
        return host.to(device, non_blocking=True)
    if device.index is not None and device.index != torch.cuda.current_device():
        raise RuntimeError(f'upload: target cuda:{device.index} is not the current device cuda:{torch.cuda.current_device()}')
    host = host.contiguous()
    out = torch.empty(host.shape, dtype=host.dtype, device=device)
    check(lib().spg_upload(host.data_ptr(), host.numel() * host.element_size(), out.data_ptr(), _stream()), 'spg_upload')
    return out


def upload_packed(hosts, device=None):
    """Several small host tensors -> ONE staging copy -> views of one device buffer (include/spg_hip.h: spg_upload_packed), in the
    order given; None entries stay None.  What a fresh batch needs on the device besides its clouds travels this way (edge list,
    edge features, CloudEmbedder's index vectors, labels, diameters): one memcpy-and-enqueue instead of one per vector."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is not None and device.index != torch.cuda.current_device():
        raise RuntimeError(f'upload_packed: target cuda:{device.index} is not the current device cuda:{torch.cuda.current_device()}')
    live = [(i, t.contiguous()) for i, t in enumerate(hosts) if t is not None]
    for _, t in live:
        if t.is_cuda:
            raise TypeError('upload_packed takes host tensors')
    n = len(live)
    out = [None] * len(hosts)
    if n == 0:
        return out
    ptrs, sizes, offs = (ctypes.c_void_p * n)(), (ctypes.c_size_t * n)(), (ctypes.c_size_t * n)()
    total = 0
    for k, (_, t) in enumerate(live):
        nb = t.numel() * t.element_size()
        ptrs[k], sizes[k], offs[k] = t.data_ptr() if nb else None, nb, total
        total += (nb + 255) & ~255
    buf = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    check(lib().spg_upload_packed(ptrs, sizes, offs, n, buf.data_ptr(), total, _stream()), 'spg_upload_packed')
    for k, (i, t) in enumerate(live):
        out[i] = buf[offs[k]:offs[k] + sizes[k]].view(t.dtype).view(t.shape)
    return out


# --------------------------------------------------------------------------------------------------
# graph structure
# --------------------------------------------------------------------------------------------------
class DeviceGraph:
    """CSR-by-target + reverse CSR built on the device from GraphConvInfo's (idxn, degs)."""

    def __init__(self, idxn: torch.Tensor, degs: torch.Tensor, n_src: Optional[int] = None):
        """n_src: number of rows of the input feature matrix (>= number of output nodes; equal for superpoint graphs)."""
        _req(idxn, torch.int64, 'idxn'); _req(degs, torch.int64, 'degs')
        self.N, self.E = int(degs.numel()), int(idxn.numel())
        self.n_src = max(self.N, int(n_src) if n_src is not None else self.N)
        nbytes = lib().spg_graph_workspace_bytes(self.N, self.n_src, self.E)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=idxn.device)
        self.idxn, self.degs = idxn, degs
        check(lib().spg_graph_build(_ptr(idxn) if self.E else None, _ptr(degs), self.N, self.n_src, self.E, _ptr(self.ws),
                                    _stream()), 'spg_graph_build')

    @classmethod
    def from_workspace(cls, ws: torch.Tensor, idxn: torch.Tensor, degs: torch.Tensor):
        """A graph whose workspace was already filled on the device (ops.batch_graph_build)."""
        g = cls.__new__(cls)
        g.N, g.E = int(degs.numel()), int(idxn.numel())
        g.n_src = g.N
        g.ws, g.idxn, g.degs = ws, idxn, degs
        return g

    def export(self):
        dev = self.ws.device
        rowptr = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
        rev_rowptr = torch.empty(self.n_src + 1, dtype=torch.int32, device=dev)
        src = torch.empty(self.E, dtype=torch.int32, device=dev)
        dst = torch.empty(self.E, dtype=torch.int32, device=dev)
        rev = torch.empty(self.E, dtype=torch.int32, device=dev)
        self.hdr = torch.empty(4, dtype=torch.int32, device=dev)
        check(lib().spg_graph_export(_ptr(self.ws), self.N, self.n_src, self.E, _ptr(rowptr), _ptr(src), _ptr(dst),
                                     _ptr(rev_rowptr), _ptr(rev), _ptr(self.hdr), _stream()), 'spg_graph_export')
        return rowptr, src, dst, rev_rowptr, rev


# --------------------------------------------------------------------------------------------------
# generic ECC operator
# --------------------------------------------------------------------------------------------------
_DT = {torch.float32: 0, torch.float64: 1}


def ecc_aggregate_fwd(x, w, graph: DeviceGraph, idxe=None, cin=None, cout=None):
    _req(x, name='input'); _req(w, x.dtype, 'weights')
    if x.shape[0] > graph.n_src:
        raise ValueError(f'graph was built for {graph.n_src} input rows, input has {x.shape[0]}')
    matrix = w.dim() == 3
    cin = x.shape[1] if cin is None else cin
    cout = (w.shape[2] if matrix else w.shape[1]) if cout is None else cout
    out = torch.empty(graph.N, cout, dtype=x.dtype, device=x.device)
    check(lib().spg_ecc_aggregate_fwd(_DT[x.dtype], _ptr(x), _ptr(w), _ptr(idxe), _ptr(graph.ws), graph.N, graph.E, cin, cout,
                                      int(matrix), _ptr(out), _stream()), 'spg_ecc_aggregate_fwd')
    return out


def ecc_aggregate_bwd(x, w, grad_out, graph: DeviceGraph, idxe=None, need_x=True, need_w=True):
    matrix = w.dim() == 3
    cin = x.shape[1]
    cout = w.shape[2] if matrix else w.shape[1]
    grad_out = grad_out.contiguous()
    gx = torch.empty_like(x) if need_x else None
    gw = torch.empty_like(w) if need_w else None
    check(lib().spg_ecc_aggregate_bwd(_DT[x.dtype], _ptr(x), _ptr(w), _ptr(idxe), _ptr(graph.ws), graph.N, graph.E,
                                      x.shape[0], w.shape[0], cin, cout, int(matrix), _ptr(grad_out), _ptr(gx), _ptr(gw),
                                      _stream()), 'spg_ecc_aggregate_bwd')
    return gx, gw


# --------------------------------------------------------------------------------------------------
# GRU cell
# --------------------------------------------------------------------------------------------------
def gru_cell_fwd(inp, hidden, params: Sequence[Optional[torch.Tensor]], layernorm: bool, ingate: bool):
    _req(inp, torch.float32, 'input'); _req(hidden, torch.float32, 'hidden')
    n = inp.shape[0]
    out = torch.empty_like(hidden)
    scratch = torch.empty(lib().spg_gru_scratch_floats(n), dtype=torch.float32, device=inp.device)
    check(lib().spg_gru_cell_fwd(_ptr(inp), _ptr(hidden), n, _ptr_array(params), int(layernorm), int(ingate), _ptr(out),
                                 _ptr(scratch), _stream()), 'spg_gru_cell_fwd')
    return out


def gru_cell_bwd(inp, hidden, grad_out, params, layernorm: bool, ingate: bool):
    n = inp.shape[0]
    grad_out = grad_out.contiguous()
    gi, gh = torch.empty_like(inp), torch.empty_like(hidden)
    grads = [None if p is None else torch.empty_like(p) for p in params]
    scratch = torch.empty(lib().spg_gru_scratch_floats(n), dtype=torch.float32, device=inp.device)
    check(lib().spg_gru_cell_bwd(_ptr(inp), _ptr(hidden), _ptr(grad_out), n, _ptr_array(params), int(layernorm), int(ingate),
                                 _ptr(gi), _ptr(gh), _ptr_array(grads), _ptr(scratch), _stream()), 'spg_gru_cell_bwd')
    return gi, gh, grads


def lstm_cell_fwd(inp, h, c, params: Sequence[Optional[torch.Tensor]], layernorm: bool, ingate: bool):
    _req(inp, torch.float32, 'input'); _req(h, torch.float32, 'hidden[0]'); _req(c, torch.float32, 'hidden[1]')
    n = inp.shape[0]
    hy, cy = torch.empty_like(h), torch.empty_like(c)
    check(lib().spg_lstm_cell_fwd(_ptr(inp), _ptr(h), _ptr(c), n, _ptr_array(params), int(layernorm), int(ingate), _ptr(hy),
                                  _ptr(cy), None, _stream()), 'spg_lstm_cell_fwd')
    return hy, cy


def lstm_cell_bwd(inp, h, c, grad_hy, grad_cy, params, layernorm: bool, ingate: bool):
    n = inp.shape[0]
    grad_hy = None if grad_hy is None else grad_hy.contiguous()
    grad_cy = None if grad_cy is None else grad_cy.contiguous()
    gi, gh, gc = torch.empty_like(inp), torch.empty_like(h), torch.empty_like(c)
    grads = [None if p is None else torch.empty_like(p) for p in params]
    scratch = torch.empty(lib().spg_lstm_scratch_floats(n), dtype=torch.float32, device=inp.device)
    check(lib().spg_lstm_cell_bwd(_ptr(inp), _ptr(h), _ptr(c), _ptr(grad_hy), _ptr(grad_cy), n, _ptr_array(params),
                                  int(layernorm), int(ingate), _ptr(gi), _ptr(gh), _ptr(gc), _ptr_array(grads), _ptr(scratch),
                                  _stream()), 'spg_lstm_cell_bwd')
    return gi, gh, gc, grads


# --------------------------------------------------------------------------------------------------
# dense layer
# --------------------------------------------------------------------------------------------------
def linear_fwd(x, w, bias=None, in_scale=None, in_shift=None, in_relu=False):
    _req(x, torch.float32, 'x'); _req(w, torch.float32, 'w')
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    check(lib().spg_linear_fwd(_ptr(x), K, M, K, _ptr(w), _ptr(bias), N, _ptr(in_scale), _ptr(in_shift), int(in_relu), _ptr(y),
                               N, _stream()), 'spg_linear_fwd')
    return y


def linear_dgrad(dy, w):
    """dx [M, K] = dy [M, N] @ w [N, K]."""
    _req(dy, torch.float32, 'dy'); _req(w, torch.float32, 'w')
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    check(lib().spg_linear_dgrad(_ptr(dy), N, M, N, _ptr(w), K, _ptr(dx), K, _stream()), 'spg_linear_dgrad')
    return dx


def colsum(x, out=None):
    _req(x, torch.float32, 'x')
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device) if out is None else out
    work = torch.empty(64 * N, dtype=torch.float32, device=x.device)
    check(lib().spg_colsum(_ptr(x), N, M, N, _ptr(out), _ptr(work), _stream()), 'spg_colsum')
    return out


def linear_wgrad(dy, x, in_scale=None, in_shift=None, in_relu=False, out=None):
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=x.device) if out is None else out
    work = torch.empty(max(1, lib().spg_linear_wgrad_work_floats(M, N, K)), dtype=torch.float32, device=x.device)
    check(lib().spg_linear_wgrad(_ptr(dy), N, _ptr(x), K, M, N, K, _ptr(in_scale), _ptr(in_shift), int(in_relu), _ptr(dw),
                                 _ptr(work), _stream()), 'spg_linear_wgrad')
    return dw


def linear_wgrad_bias(dy, x, out_w=None, out_b=None):
    """dW [N, K] and dbias [N] of a dense layer in two launches (spg_linear_wgrad_bias)."""
    _req(dy, torch.float32, 'dy'); _req(x, torch.float32, 'x')
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=x.device) if out_w is None else out_w
    db = torch.empty(N, dtype=torch.float32, device=x.device) if out_b is None else out_b
    work = torch.empty(max(1, lib().spg_linear_wgrad_bias_work_floats(M, N, K)), dtype=torch.float32, device=x.device)
    check(lib().spg_linear_wgrad_bias(_ptr(dy), N, _ptr(x), K, M, N, K, None, None, 0, _ptr(dw), _ptr(db), _ptr(work), _stream()),
          'spg_linear_wgrad_bias')
    return dw, db


def linear_backward(dy, x, w, need_dx=True, has_bias=True, out_w=None, out_b=None):
    """The whole backward of y = x w^T + b: (dx or None, dW, dbias or None) -- ONE grouped launch + one batched reduction
    (spg_linear_backward) instead of three launches."""
    _req(dy, torch.float32, 'dy'); _req(x, torch.float32, 'x'); _req(w, torch.float32, 'w')
    M, N = dy.shape
    K = x.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=x.device) if need_dx else None
    dw = torch.empty(N, K, dtype=torch.float32, device=x.device) if out_w is None else out_w
    db = (torch.empty(N, dtype=torch.float32, device=x.device) if out_b is None else out_b) if has_bias else None
    work = torch.empty(max(1, lib().spg_linear_wgrad_bias_work_floats(M, N, K)), dtype=torch.float32, device=x.device)
    check(lib().spg_linear_backward(_ptr(dy), N, _ptr(x), K, _ptr(w), M, N, K, _ptr(dx), K, _ptr(dw), _ptr(db), _ptr(work), _stream()),
          'spg_linear_backward')
    return dx, dw, db


# --------------------------------------------------------------------------------------------------
# PointNet
# --------------------------------------------------------------------------------------------------
def make_pointnet_cfg(nfeat, nfeat_stn, nfeat_global, npts, stn_conv, stn_fc, conv, fc, last_ac=False,
                      bn_eps=1e-5, bn_momentum=0.1) -> PointNetCfg:
    c = PointNetCfg()
    c.nfeat, c.nfeat_stn, c.nfeat_global, c.npts = nfeat, nfeat_stn, nfeat_global, npts
    c.n_stn_conv, c.n_stn_fc, c.n_conv, c.n_fc = len(stn_conv), len(stn_fc), len(conv), len(fc)
    for dst, src in ((c.stn_conv, stn_conv), (c.stn_fc, stn_fc), (c.conv, conv), (c.fc, fc)):
        if len(src) > _lib.SPG_MAX_LAYERS:
            raise ValueError('too many layers')
        for i, v in enumerate(src):
            dst[i] = int(v)
    c.last_ac, c.bn_eps, c.bn_momentum = int(last_ac), bn_eps, bn_momentum
    return c


def _require_training_state(state, what):
    """The backward kernels read the batch-statistics workspace layout of a TRAINING-mode forward (saved aggregates,
    BatchNorm mean / rstd).  After an eval-mode forward that layout does not exist: fail loudly instead of reading
    out of bounds (frozen-BatchNorm fine-tuning is not implemented on the HIP path)."""
    if not state.training:
        raise RuntimeError(f'{what}: backward() after an eval-mode forward is not supported by the HIP path '
                           '(the backward implements batch-statistics BatchNorm); call model.train() before the forward, '
                           'or wrap the eval-mode forward in torch.no_grad()')


class PointNetState:
    """Saved forward state (workspace with the raw layer outputs and BatchNorm constants)."""

    def __init__(self, cfg, B, clouds, clouds_global, ws, training, ext_transform=None):
        self.cfg, self.B, self.clouds, self.clouds_global, self.ws = cfg, B, clouds, clouds_global, ws
        self.training = bool(training)
        self.ext_transform = ext_transform


def pointnet_forward(cfg: PointNetCfg, clouds, clouds_global, groups: List[Sequence[Optional[torch.Tensor]]],
                     training: bool, bn_update_times: int = 1, ext_transform=None):
    """groups: one 6-tuple (weight, bias, bn.weight, bn.bias, running_mean, running_var) per layer in the order
    of include/spg_hip.h.  ext_transform: [B, 4] = T - I of an externally evaluated STN (PointNet without inner STN).
    Returns (emb [B, D], PointNetState)."""
    _req(clouds, torch.float32, 'clouds')
    B = clouds.shape[0]
    if clouds.shape[1] != cfg.nfeat or clouds.shape[2] != cfg.npts:
        raise ValueError(f'clouds must be [B, {cfg.nfeat}, {cfg.npts}], got {tuple(clouds.shape)}')
    if training and B <= 1:
        # torch.nn.BatchNorm1d raises for the FC layers ([1, C] / empty input) in training mode; so do we
        raise ValueError(f'Expected more than 1 value per channel when training, got input size [{B}, C]')
    if B == 0:      # inference on a batch without a single embeddable superpoint: nothing to launch
        return (torch.zeros(0, cfg.fc[cfg.n_fc - 1], dtype=torch.float32, device=clouds.device),
                PointNetState(cfg, 0, clouds, clouds_global, None, training, ext_transform))
    if clouds_global is not None:
        clouds_global = _req(clouds_global.reshape(B, -1).contiguous(), torch.float32, 'clouds_global')
        if clouds_global.shape[1] != cfg.nfeat_global:
            raise ValueError('clouds_global width does not match nfeat_global')
    nbytes = lib().spg_pointnet_workspace_bytes(ctypes.byref(cfg), B, int(training))
    if nbytes == 0:
        raise RuntimeError('spg_pointnet_workspace_bytes: ' + lib().spg_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=clouds.device)
    D = cfg.fc[cfg.n_fc - 1]
    emb = torch.empty(B, D, dtype=torch.float32, device=clouds.device)
    flat = [t for g in groups for t in g]
    if ext_transform is not None:
        ext_transform = _req(ext_transform.reshape(B, 4).contiguous(), torch.float32, 'ext_transform')
    check(lib().spg_pointnet_forward_ext(ctypes.byref(cfg), B, _ptr(clouds), _ptr(clouds_global), _ptr(ext_transform), _ptr_array(flat),
                                         _ptr(emb), _ptr(ws), int(training), int(bn_update_times), _stream()), 'spg_pointnet_forward')
    return emb, PointNetState(cfg, B, clouds, clouds_global, ws, training, ext_transform)


def pointnet_backward(state: PointNetState, groups, grad_emb, out_grads=None, want_input_grads=False):
    """Returns a list of 4-tuples (d weight, d bias, d bn.weight, d bn.bias) per layer.
    out_grads: optional pre-allocated destination tensors in the same structure (written, not accumulated); a
    `None` entry is skipped by the library (used for the analytically-zero biases in front of a BatchNorm when the
    destination is a pre-zeroed flat gradient buffer)."""
    cfg, B = state.cfg, state.B
    _require_training_state(state, 'PointNet')
    if getattr(state, 'backward_done', False):
        # the BatchNorm-backward sums are accumulated into slots of the forward workspace that the forward cleared (spg_gemm.h):
        # a second backward pass over the same forward would add to them again
        raise RuntimeError('PointNet: a second backward() through the same forward is not supported by the HIP path '
                           '(run the forward again)')
    state.backward_done = True
    grad_emb = _req(grad_emb.contiguous(), torch.float32, 'grad_emb')
    nbytes = lib().spg_pointnet_bwd_workspace_bytes(ctypes.byref(cfg), B)
    bws = torch.empty(nbytes, dtype=torch.uint8, device=grad_emb.device)
    gg, flatg = [], []
    for li, g in enumerate(groups):
        if out_grads is not None:
            d = list(out_grads[li][:4])
        else:
            d = [None if g[k] is None else torch.empty_like(g[k]) for k in range(4)]
        gg.append(tuple(d))
        flatg += d + [None, None]
    flat = [t for g in groups for t in g]
    g_T = g_glob = None
    if want_input_grads:          # gradients wrt the external transform and the global features (LocalCloudEmbedder)
        if state.ext_transform is not None:
            g_T = torch.empty(B, 4, dtype=torch.float32, device=grad_emb.device)
        if state.clouds_global is not None:
            g_glob = torch.empty_like(state.clouds_global)
    check(lib().spg_pointnet_backward_ext(ctypes.byref(cfg), B, _ptr(state.clouds), _ptr(state.clouds_global), _ptr(state.ext_transform),
                                          _ptr_array(flat), _ptr(grad_emb), _ptr_array(flatg), _ptr(g_T), _ptr(g_glob), _ptr(state.ws),
                                          _ptr(bws), _stream()), 'spg_pointnet_backward')
    if want_input_grads:
        return gg, g_T, g_glob
    return gg


# --------------------------------------------------------------------------------------------------
# RNN-ECC module
# --------------------------------------------------------------------------------------------------
def make_eccrnn_cfg(nc, nrepeats, matrix, layernorm, ingate, cat_all, fnet_widths, bnidx, llbias, bn_eps=1e-5,
                    bn_momentum=0.1, cell='gru') -> EccRnnCfg:
    c = EccRnnCfg()
    c.nc, c.nrepeats, c.matrix, c.layernorm, c.ingate, c.cat_all = nc, nrepeats, int(matrix), int(layernorm), int(ingate), int(cat_all)
    c.n_fnet = len(fnet_widths) - 1
    if c.n_fnet > _lib.SPG_MAX_LAYERS:
        raise ValueError('filter network too deep')
    for i, v in enumerate(fnet_widths):
        c.fnet_widths[i] = int(v)
    c.bnidx, c.llbias, c.bn_eps, c.bn_momentum = bnidx, int(llbias), bn_eps, bn_momentum
    c.cell = {'gru': 0, 'lstm': 1}[cell]
    return c


class EccRnnState:
    def __init__(self, cfg, graph, edgefeats, ws, training):
        self.cfg, self.graph, self.edgefeats, self.ws = cfg, graph, edgefeats, ws
        self.training = bool(training)


def eccrnn_forward(cfg: EccRnnCfg, graph: DeviceGraph, h0, edgefeats, groups, training: bool, bn_update_times: int = 1):
    _req(h0, torch.float32, 'input'); _req(edgefeats, torch.float32, 'edgefeats')
    N, E = graph.N, graph.E
    if h0.shape[0] != N or h0.shape[1] != cfg.nc:
        raise ValueError(f'input must be [{N}, {cfg.nc}], got {tuple(h0.shape)}')
    if edgefeats.dim() != 2 or edgefeats.shape[0] != E or edgefeats.shape[1] != cfg.fnet_widths[0]:
        raise ValueError(f'edgefeats must be [{E}, {cfg.fnet_widths[0]}], got {tuple(edgefeats.shape)}')
    if training and cfg.bnidx >= 0 and E == 1:
        raise ValueError('Expected more than 1 value per channel when training (filter-network BatchNorm over one edge)')
    nbytes = lib().spg_eccrnn_workspace_bytes(ctypes.byref(cfg), N, E, int(training))
    if nbytes == 0:
        raise RuntimeError('spg_eccrnn_workspace_bytes: ' + lib().spg_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h0.device)
    width = cfg.nc * (cfg.nrepeats + 1) if cfg.cat_all else cfg.nc
    out = torch.empty(N, width, dtype=torch.float32, device=h0.device)
    flat = [t for g in groups for t in g]
    check(lib().spg_eccrnn_forward(ctypes.byref(cfg), N, E, _ptr(graph.ws), _ptr(h0), _ptr(edgefeats), _ptr_array(flat),
                                   _ptr(out), _ptr(ws), int(training), int(bn_update_times), _stream()), 'spg_eccrnn_forward')
    return out, EccRnnState(cfg, graph, edgefeats, ws, training)


def eccrnn_backward(state: EccRnnState, groups, grad_out, out_grads=None):
    """out_grads: see pointnet_backward."""
    cfg, graph = state.cfg, state.graph
    N, E = graph.N, graph.E
    _require_training_state(state, 'RNN-ECC')
    grad_out = _req(grad_out.contiguous(), torch.float32, 'grad_out')
    bws = torch.empty(lib().spg_eccrnn_bwd_workspace_bytes(ctypes.byref(cfg), N, E), dtype=torch.uint8, device=grad_out.device)
    grad_h0 = torch.empty(N, cfg.nc, dtype=torch.float32, device=grad_out.device)
    gg, flatg = [], []
    nf = cfg.n_fnet
    for li, g in enumerate(groups):
        if out_grads is not None:
            d = (list(out_grads[li][:4]) + [None, None]) if li < nf else list(out_grads[li][:6])
        elif li < nf:
            d = [None if g[k] is None else torch.empty_like(g[k]) for k in range(4)] + [None, None]
        else:
            d = [None if g[k] is None else torch.empty_like(g[k]) for k in range(6)]
        gg.append(tuple(d))
        flatg += d
    flat = [t for g in groups for t in g]
    check(lib().spg_eccrnn_backward(ctypes.byref(cfg), N, E, _ptr(graph.ws), _ptr(state.edgefeats), _ptr_array(flat),
                                    _ptr(grad_out), _ptr(grad_h0), _ptr_array(flatg), _ptr(state.ws), _ptr(bws), _stream()),
          'spg_eccrnn_backward')
    return grad_h0, gg


# --------------------------------------------------------------------------------------------------
# batch construction on the device
# --------------------------------------------------------------------------------------------------
def persistent_ecc_status(clear=False):
    """(time-outs, withheld optimiser updates) of the dataflow-synchronised RNN-ECC launches of the current device
    (include/spg_hip.h: spg_ecc_persistent_status).  The time-out word is sticky and read by the fused clamp + Adam launch itself: while
    it is non-zero every update is WITHHELD (parameters / moments bit-identical), so nothing computed from stale neighbour states
    reaches the model before the host has looked.  One blocking 16-byte copy: synchronises the device."""
    e, w = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().spg_ecc_persistent_status(ctypes.byref(e), ctypes.byref(w), 1 if clear else 0), 'spg_ecc_persistent_status')
    return int(e.value), int(w.value)


def recover_persistent_ecc(arena=None, group=None):
    """The per-step fail-safe's host half: reads and clears the status; when a recurrence has timed out, switches this process (every
    rank of `group`: the decision is all-reduced) to the per-iteration kernels (spg_tune key 8, which cannot time out) and takes the
    withheld updates off `arena`'s Adam step counter (FlatParameters.rewind_steps).  Returns the number of withheld updates (0:
    nothing happened) -- the caller repeats the batches it still holds.  COLLECTIVE when a process group is initialised (every rank
    must call it at the same step; group=False keeps it local)."""
    errors, withheld = persistent_ecc_status(clear=True)
    total = errors
    if group is not False:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            on_gpu = dist.get_backend(group) == 'nccl'
            t = torch.tensor([float(errors)], dtype=torch.float64, device=torch.device('cuda', torch.cuda.current_device()) if on_gpu else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total = int(t.item())
    if total == 0:
        return 0
    lib().spg_tune(8, 1)
    if arena is not None and withheld:
        arena.rewind_steps(withheld)
    return max(withheld, 1)


def check_persistent_ecc(what='training', group=None):
    """Health check of the dataflow-synchronised RNN-ECC launches (include/spg_hip.h: spg_ecc_persistent_status): a
    wave whose bounded spin ran out (a peer workgroup was not resident in time: shared GPU, profiler, another stream holding
    CUs) carried on with stale neighbour states.  Since round 6 the fused optimiser launch WITHHOLDS its update while the time-out word
    is set, so the parameters are never touched by such a step; this check is the last line (epoch end, before a checkpoint).
    Synchronises the device.  Raises after clearing the counters and switching this process to the per-iteration kernels
    (spg_tune key 8), so a caller that catches the error can repeat the affected work safely.
    COLLECTIVE CONTRACT: with an initialised process group the count is summed over the ranks of `group` first -- EVERY rank of the
    group must call it at the same point (a read failure on one rank is folded into the reduced value and raised on every rank, so no
    rank is left alone in the collective); group=False keeps it local."""
    failed = None
    try:
        n, withheld = persistent_ecc_status(clear=True)
    except RuntimeError as exc:      # the counter could not be read: a different failure, not "-1 time-outs" -- but the collective still runs
        failed, n, withheld = exc, 0, 0
    total, any_failed = n, failed is not None
    if group is not False:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            # data parallel: every rank must take the same decision (one rank raising alone leaves the others in the next
            # collective until the RCCL time-out, and the ranks would run different kernels afterwards)
            on_gpu = dist.get_backend(group) == 'nccl'
            t = torch.tensor([float(n), 1.0 if failed is not None else 0.0], dtype=torch.float64,
                             device=torch.device('cuda', torch.cuda.current_device()) if on_gpu else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total, any_failed = int(t[0].item()), bool(t[1].item() > 0)
    if any_failed:
        raise RuntimeError(f'the persistent RNN-ECC status could not be read during {what} on '
                           + ('this rank: ' + str(failed) if failed is not None else 'another rank of the job'))
    if total == 0:
        return
    lib().spg_tune(8, 1)
    raise RuntimeError(f'{total} persistent RNN-ECC spin time-out(s) ({n} on this rank; {withheld} optimiser update(s) withheld here) during '
                       f'{what}: the ECC outputs / gradients of the affected steps are wrong (a workgroup of the dataflow-synchronised launch '
                       'was not resident in time); the fused optimiser step withheld its updates from the first time-out on.  The process '
                       '(every rank of the job) now uses the per-iteration kernels (spg_tune key 8); repeat the work since the last check')


def set_batch(edges, n_nodes: int):
    """edges i64 [E, 2] (source, target; batch node offsets applied) on the device -> (idxn i64 [E], degs i64 [N],
    perm i64 [E]): edges ordered by target (stable), see include/spg_hip.h."""
    _req(edges, torch.int64, 'edges')
    E = int(edges.shape[0])
    dev = edges.device
    idxn = torch.empty(E, dtype=torch.int64, device=dev)
    perm = torch.empty(E, dtype=torch.int64, device=dev)
    degs = torch.empty(n_nodes, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(lib().spg_set_batch_workspace_bytes(n_nodes, E), dtype=torch.uint8, device=dev)
    check(lib().spg_set_batch(_ptr(edges) if E else None, n_nodes, E, _ptr(idxn) if E else None, _ptr(degs), _ptr(perm) if E else None,
                              _ptr(ws), _ptr(err), _stream()), 'spg_set_batch')
    return idxn, degs, perm, err


def batch_graph_fits(n_nodes: int, n_edges: int, n_feat: int) -> bool:
    """True when the single-launch batch builder serves a batch of this size (include/spg_hip.h: spg_batch_graph_scratch_bytes)."""
    return n_edges > 0 and lib().spg_batch_graph_scratch_bytes(int(n_nodes), int(n_edges), int(n_feat)) > 0


def batch_graph_build(edges_h: torch.Tensor, feats_h: Optional[torch.Tensor], n_nodes: int, device=None):
    """The whole construction of a small batch in one launch (include/spg_hip.h: spg_batch_graph_build / _dev): edges i64 [E, 2]
    (batch node offsets applied) and edge features f32 [E, F], both on the HOST (uploaded here) or both on the DEVICE already -> (idxn, degs, feats_sorted, DeviceGraph, error flag) on the
    device, or None when the batch is too large for the single-workgroup builder (use set_batch + gather_rows + DeviceGraph)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    E = int(edges_h.shape[0])
    F = 0 if feats_h is None else int(feats_h.shape[1])
    L = lib()
    nscratch = L.spg_batch_graph_scratch_bytes(n_nodes, E, F)
    on_dev = edges_h.is_cuda
    if nscratch == 0 or (feats_h is not None and feats_h.is_cuda != on_dev):
        return None
    edges_h = edges_h.contiguous()
    if edges_h.dtype != torch.int64:
        raise TypeError('edges must be int64')
    if feats_h is not None:
        feats_h = feats_h.contiguous()
        if feats_h.dtype != torch.float32 or feats_h.shape[0] != E:
            raise TypeError(f'edge features must be float32 [{E}, F]')
    idxn = torch.empty(E, dtype=torch.int64, device=device)
    degs = torch.empty(n_nodes, dtype=torch.int64, device=device)
    feats = torch.empty(E, F, dtype=torch.float32, device=device) if feats_h is not None else None
    ws = torch.empty(L.spg_graph_workspace_bytes(n_nodes, n_nodes, E), dtype=torch.uint8, device=device)
    scratch = torch.empty(nscratch, dtype=torch.uint8, device=device)
    err = torch.empty(1, dtype=torch.int32, device=device)
    # (inputs already on the device: uploaded with the batch's other small vectors by ONE upload_packed)
    check((L.spg_batch_graph_build_dev if on_dev else L.spg_batch_graph_build)(edges_h.data_ptr() if E else None, feats_h.data_ptr() if (feats_h is not None and E) else None, n_nodes, E, F,
                                  _ptr(idxn) if E else None, _ptr(degs), _ptr(feats) if (feats is not None and E) else None, _ptr(ws),
                                  _ptr(scratch), _ptr(err), _stream()), 'spg_batch_graph_build')
    return idxn, degs, feats, DeviceGraph.from_workspace(ws, idxn, degs), err


def gather_rows(src, perm):
    _req(src, torch.float32, 'src'); _req(perm, torch.int64, 'perm')
    rows, cols = int(perm.numel()), int(src.shape[1])
    dst = torch.empty(rows, cols, dtype=torch.float32, device=src.device)
    check(lib().spg_gather_rows(_ptr(src), cols, _ptr(perm), rows, cols, _ptr(dst), cols, _stream()), 'spg_gather_rows')
    return dst


_EF_KIND = {'': 0, 'copy': 0, 'd': 1, 'ld': 2, 'r': 3, 'const': 4}


def edge_features(columns, edges, mean=None, scale=None):
    """columns: list of (kind, tensor or None, column) per OUTPUT column with kind in copy|d|ld|r|const; the tensor is a
    float32 / float64 device matrix (per edge for copy, per node otherwise).  edges i64 [E, 2]; mean / scale f64 [ncols]."""
    _req(edges, torch.int64, 'edges')
    specs = _lib.EdgeFeatureSpecs()
    if len(columns) > _lib.SPG_EF_MAX_COLS:
        raise ValueError('too many edge feature columns')
    specs.ncols = len(columns)
    keep = []
    for i, (kind, t, col) in enumerate(columns):
        sp = specs.col[i]
        sp.kind = _EF_KIND[kind]
        if t is not None:
            _req(t, name='attribute')
            if t.dtype not in (torch.float32, torch.float64) or t.dim() != 2:
                raise TypeError('attributes must be 2-d float32 / float64 matrices')
            sp.data, sp.ld, sp.column, sp.is_f64 = t.data_ptr(), t.shape[1], int(col), int(t.dtype == torch.float64)
            keep.append(t)
    E = int(edges.shape[0])
    out = torch.empty(E, len(columns), dtype=torch.float32, device=edges.device)
    check(lib().spg_edge_features(ctypes.byref(specs), _ptr(edges) if E else None, E, _ptr(mean), _ptr(scale), _ptr(out), _stream()),
          'spg_edge_features')
    return out


# --------------------------------------------------------------------------------------------------
# superpoint loader
# --------------------------------------------------------------------------------------------------
def load_superpoints(points, offsets, slot, sample_idx, colmap, xyznormalize: bool, n_valid: int, M=None, noise=None):
    """points f32 [Ntot, ncols], offsets i64 [S+1], slot i32 [S], sample_idx i32 [S, npts] (on the device), colmap:
    sequence of raw column indices (host) -> clouds f32 [n_valid, F, npts], diam f32 [n_valid].
    M: f64 [S, 3, 3] or None, noise: f32 [n_valid, npts, F] or None (device)."""
    _req(points, torch.float32, 'points'); _req(offsets, torch.int64, 'offsets'); _req(slot, torch.int32, 'slot')
    _req(sample_idx, torch.int32, 'sample_idx')
    colmap_h = (ctypes.c_int32 * len(colmap))(*[int(c) for c in colmap])
    S, npts, F = slot.numel(), sample_idx.shape[1], len(colmap)
    if offsets.numel() != S + 1 or sample_idx.shape[0] != S:
        raise ValueError('offsets / slot / sample_idx disagree on the number of superpoints')
    if M is not None:
        _req(M, torch.float64, 'M')
    if noise is not None:
        _req(noise, torch.float32, 'noise')
    clouds = torch.empty(n_valid, F, npts, dtype=torch.float32, device=points.device)
    diam = torch.empty(n_valid, dtype=torch.float32, device=points.device)
    check(lib().spg_load_superpoints(_ptr(points), points.shape[1], _ptr(offsets), S, _ptr(slot), _ptr(sample_idx), npts,
                                     int(bool(xyznormalize)), ctypes.cast(colmap_h, ctypes.c_void_p), F, _ptr(M), _ptr(noise), _ptr(clouds),
                                     _ptr(diam), _stream()), 'spg_load_superpoints')
    return clouds, diam


def loader_random(counts, ids, slot, npts: int, nfeat: int, n_valid: int, seed: int, step: int, augment: bool,
                  scale: float = 0.0, rot: bool = False, mirror_prob: float = 0.0, jitter: bool = False):
    """The loader's random streams generated on the device (Philox4x32-10 keyed by (seed, superpoint id, step)):
    counts / ids i64 [S], slot i32 [S] -> sample_idx i32 [S, npts], M f64 [S, 3, 3] or None, noise f32 [n_valid, npts, nfeat]
    or None -- the inputs of `load_superpoints` (reference learning/spg.py:207-214, 241-257)."""
    _req(counts, torch.int64, 'counts'); _req(ids, torch.int64, 'ids'); _req(slot, torch.int32, 'slot')
    S = slot.numel()
    if counts.numel() != S or ids.numel() != S:
        raise ValueError('counts / ids / slot disagree on the number of superpoints')
    dev = counts.device
    sidx = torch.empty(S, npts, dtype=torch.int32, device=dev)
    M = torch.empty(S, 3, 3, dtype=torch.float64, device=dev) if augment else None
    noise = torch.empty(n_valid, npts, nfeat, dtype=torch.float32, device=dev) if (augment and jitter) else None
    check(lib().spg_loader_random(_ptr(counts), _ptr(ids), _ptr(slot), S, int(npts), int(nfeat), int(seed) & (2 ** 64 - 1),
                                  int(step) & 0xFFFFFFFF, int(bool(augment)), float(scale), int(bool(rot)), float(mirror_prob),
                                  int(bool(jitter)), _ptr(sidx), _ptr(M), _ptr(noise), _stream()), 'spg_loader_random')
    return sidx, M, noise


# --------------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------------
class _CrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index, mean):
        logits = _req(logits.contiguous(), torch.float32, 'logits')
        target = _req(target.contiguous(), torch.int64, 'target')
        N, C = logits.shape
        buf = torch.empty(N + 2, dtype=torch.float32, device=logits.device)          # lse [N] | loss | normaliser
        check(lib().spg_cross_entropy_fwd(_ptr(logits), _ptr(target), _ptr(weight), N, C, int(ignore_index), int(mean),
                                          buf[N:].data_ptr(), _ptr(buf), buf[N + 1:].data_ptr(), _stream()), 'spg_cross_entropy_fwd')
        ctx.save_for_backward(logits, target, buf)
        ctx.weight, ctx.ignore_index, ctx.mean = weight, int(ignore_index), int(mean)
        _CrossEntropyFunction.last_normaliser = buf[N + 1:N + 2]      # sum of the class weights of the labelled rows (device)
        return buf[N]

    @staticmethod
    def backward(ctx, grad_loss):
        logits, target, buf = ctx.saved_tensors
        N, C = logits.shape
        g = torch.empty_like(logits)
        grad_loss = grad_loss.contiguous().float()
        check(lib().spg_cross_entropy_bwd(_ptr(logits), _ptr(target), _ptr(ctx.weight), _ptr(buf), buf[N + 1:].data_ptr(), _ptr(grad_loss),
                                          N, C, ctx.ignore_index, ctx.mean, _ptr(g), _stream()), 'spg_cross_entropy_bwd')
        return g, None, None, None, None


def cross_entropy(logits, target, weight=None, ignore_index=-100, reduction='mean', return_normaliser=False):
    """torch.nn.functional.cross_entropy for [N, C] logits and class-index targets (the form learning/main.py:205 uses),
    forward and backward one HIP launch each.  return_normaliser: also the [1] device tensor sum_i weight[target_i] over the
    labelled rows -- the loss weight w_r of a data-parallel rank (superpoint_graph_amd/dist.py), without a host copy."""
    if reduction not in ('mean', 'sum'):
        raise NotImplementedError("reduction must be 'mean' or 'sum'")
    if logits.dim() != 2 or target.dim() != 1 or target.shape[0] != logits.shape[0]:
        raise ValueError('cross_entropy expects logits [N, C] and targets [N]')
    if weight is not None:
        weight = _req(weight.contiguous(), torch.float32, 'weight')
    loss = _CrossEntropyFunction.apply(logits, target, weight, ignore_index, reduction == 'mean')
    if return_normaliser:
        return loss, _CrossEntropyFunction.last_normaliser
    return loss


# --------------------------------------------------------------------------------------------------
# evaluation accounting
# --------------------------------------------------------------------------------------------------
def eval_accumulate(logits, label_mode, label_vec, confusion, counters):
    """logits f32 [N, C] or [S, N, C]; label_mode i64 [N]; label_vec i64 [N, C]; confusion i64 [C, C] and counters i64 [2]
    are accumulated in place.  Returns pred i64 [N]."""
    _req(logits, torch.float32, 'logits'); _req(label_mode, torch.int64, 'label_mode'); _req(label_vec, torch.int64, 'label_vec')
    _req(confusion, torch.int64, 'confusion'); _req(counters, torch.int64, 'counters')
    S = 1 if logits.dim() == 2 else logits.shape[0]
    N, C = logits.shape[-2], logits.shape[-1]
    if label_vec.shape != (N, C) or label_mode.shape != (N,) or confusion.shape != (C, C):
        raise ValueError('shapes: logits [S,]N,C; label_vec N,C; label_mode N; confusion C,C')
    pred = torch.empty(N, dtype=torch.int64, device=logits.device)
    check(lib().spg_eval_accumulate(_ptr(logits), S, N * C, N, C, _ptr(label_mode), _ptr(label_vec), _ptr(pred),
                                    _ptr(confusion), _ptr(counters), _stream()), 'spg_eval_accumulate')
    return pred


# --------------------------------------------------------------------------------------------------
# superpoint-graph construction (partition/graphs.py compute_sp_graph after the triangulation, ply_c compute_geof)
# --------------------------------------------------------------------------------------------------
def _u8_workspace(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def sp_graph(xyz, comp, n_com: int, tets, d_max: float, labels=None, label_rows=None, n_labels: int = 0):
    """xyz f32 [n,3], comp i32 [n], tets i32 [T,4] (Delaunay simplices), all on the device -> dict of device tensors with the
    keys of the reference's compute_sp_graph (partition/graphs.py:75-210).  Three host synchronisations (the numbers of raw
    interface pairs, unique edges and superedges size the next stage's buffers)."""
    _req(xyz, torch.float32, 'xyz'); _req(comp, torch.int32, 'comp'); _req(tets, torch.int32, 'tets')
    L, dev, st = lib(), xyz.device, _stream()
    n, T = int(xyz.shape[0]), int(tets.shape[0])
    if xyz.dim() != 2 or xyz.shape[1] != 3 or comp.numel() != n or (T and tets.shape[1] != 4):
        raise ValueError('sp_graph: xyz [n,3], comp [n], tets [T,4] expected')
    u64, f32 = torch.int64, torch.float32                 # (torch has no uint64 arithmetic: int64 storage, the library reads it unsigned)
    cnt = torch.zeros(1, dtype=u64, device=dev)
    # ---- superpoints (graphs.py:141-172) ----
    out = {'sp_centroids': torch.empty(n_com, 3, dtype=f32, device=dev), 'sp_length': torch.empty(n_com, 1, dtype=f32, device=dev),
           'sp_surface': torch.empty(n_com, 1, dtype=f32, device=dev), 'sp_volume': torch.empty(n_com, 1, dtype=f32, device=dev),
           'sp_point_count': torch.empty(n_com, 1, dtype=u64, device=dev)}
    sp_labels = None
    if labels is not None or label_rows is not None:
        sp_labels = torch.empty(n_com, n_labels + 1, dtype=torch.int32, device=dev)
        if labels is not None:
            _req(labels, torch.int32, 'labels')
        else:
            _req(label_rows, torch.int32, 'label_rows')
    ws = _u8_workspace(L.spg_spg_workspace_bytes(2, n), dev)
    check(L.spg_spg_superpoints(_ptr(xyz), n, _ptr(comp), n_com, _ptr(labels), _ptr(label_rows), n_labels, _ptr(out['sp_centroids']),
                                _ptr(out['sp_length']), _ptr(out['sp_surface']), _ptr(out['sp_volume']), _ptr(out['sp_point_count']),
                                _ptr(sp_labels), _ptr(ws), ws.numel(), st), 'spg_spg_superpoints')
    out['sp_labels'] = sp_labels
    # ---- interface edges of the tetrahedra, unique, shorter than d_max (:85-113) ----
    keys = torch.empty(max(12 * T, 1), dtype=u64, device=dev)
    check(L.spg_spg_tet_edges(_ptr(tets) if T else None, T, _ptr(comp), _ptr(keys), 12 * T, _ptr(cnt), st), 'spg_spg_tet_edges')
    n_raw = int(cnt.item())
    edge_keys = torch.empty(max(n_raw, 1), dtype=u64, device=dev)
    cc_keys = torch.empty(max(n_raw, 1), dtype=u64, device=dev)
    ws = _u8_workspace(L.spg_spg_workspace_bytes(0, n_raw), dev)
    check(L.spg_spg_unique_edges(_ptr(keys), n_raw, _ptr(xyz), _ptr(comp), n_com, float(d_max), _ptr(edge_keys), _ptr(cc_keys), _ptr(cnt),
                                 _ptr(ws), ws.numel(), st), 'spg_spg_unique_edges')
    n_edg = int(cnt.item())
    # ---- ordered by component pair, superedge segments (:117-128) ----
    cc_sorted = torch.empty(max(n_edg, 1), dtype=u64, device=dev)
    edges_sorted = torch.empty(max(n_edg, 1), dtype=u64, device=dev)
    seg_cc = torch.empty(max(n_edg, 1), dtype=u64, device=dev)
    seg_off = torch.empty(n_edg + 1, dtype=u64, device=dev)
    ws = _u8_workspace(L.spg_spg_workspace_bytes(1, n_edg), dev)
    check(L.spg_spg_group_edges(_ptr(cc_keys), _ptr(edge_keys), n_edg, _ptr(cc_sorted), _ptr(edges_sorted), _ptr(seg_cc), _ptr(seg_off),
                                _ptr(cnt), _ptr(ws), ws.numel(), st), 'spg_spg_group_edges')
    n_sedg = int(cnt.item())
    # ---- superedge features (:174-208) ----
    se = {'source': torch.empty(n_sedg, 1, dtype=torch.int32, device=dev), 'target': torch.empty(n_sedg, 1, dtype=torch.int32, device=dev)}
    for k, w in (('se_delta_mean', 3), ('se_delta_std', 3), ('se_delta_norm', 1), ('se_delta_centroid', 3), ('se_length_ratio', 1),
                 ('se_surface_ratio', 1), ('se_volume_ratio', 1), ('se_point_count_ratio', 1)):
        se[k] = torch.empty(n_sedg, w, dtype=f32, device=dev)
    check(L.spg_spg_superedges(_ptr(edges_sorted), _ptr(seg_cc), _ptr(seg_off), n_sedg, n_com, _ptr(xyz), _ptr(out['sp_centroids']),
                               _ptr(out['sp_length']), _ptr(out['sp_surface']), _ptr(out['sp_volume']), _ptr(out['sp_point_count']),
                               _ptr(se['source']), _ptr(se['target']), _ptr(se['se_delta_mean']), _ptr(se['se_delta_std']),
                               _ptr(se['se_delta_norm']), _ptr(se['se_delta_centroid']), _ptr(se['se_length_ratio']),
                               _ptr(se['se_surface_ratio']), _ptr(se['se_volume_ratio']), _ptr(se['se_point_count_ratio']), st),
          'spg_spg_superedges')
    out.update(se)
    out['edges'] = edges_sorted[:n_edg]                    # (source << 32 | target) of every Delaunay edge behind the superedges
    out['seg_off'] = seg_off[:n_sedg + 1]
    return out


def compute_geof(xyz, target, k_nn: int):
    """xyz f32 [n,3], target (u)int32 [n * k_nn] neighbour indices on the device -> geof f32 [n,4] (ply_c.cpp:384-462)."""
    _req(xyz, torch.float32, 'xyz'); _req(target, torch.int32, 'target')
    n = int(xyz.shape[0])
    if target.numel() != n * k_nn:
        raise ValueError(f'compute_geof: {n} points x {k_nn} neighbours need {n * k_nn} targets, got {target.numel()}')
    geof = torch.empty(n, 4, dtype=torch.float32, device=xyz.device)
    check(lib().spg_compute_geof(_ptr(xyz), _ptr(target), n, int(k_nn), _ptr(geof), _stream()), 'spg_compute_geof')
    return geof


def prune(xyz, voxel_size: float, rgb=None, labels=None, objects=None, n_labels: int = 0, n_objects: int = 0):
    """Voxel-grid subsampling (ply_c.cpp:288-382) of device tensors: xyz f32 [n,3], rgb u8 [n,3] or None, labels u8 [n] or None,
    objects i32 [n] (read as uint32) or None -> (xyz f32 [V,3], rgb u8 [V,3], labels i32 [V, n_labels+1], objects i32 [V, n_objects+1]);
    voxels in the order of their first point.  One host synchronisation (the number of voxels)."""
    _req(xyz, torch.float32, 'xyz')
    L, dev, st = lib(), xyz.device, _stream()
    n = int(xyz.shape[0])
    if rgb is not None:
        _req(rgb, torch.uint8, 'rgb')
    if labels is not None:
        _req(labels, torch.uint8, 'labels')
    if objects is not None:
        _req(objects, torch.int32, 'objects')
    ws = _u8_workspace(L.spg_prune_workspace_bytes(n), dev)
    nv = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.spg_prune_voxels(_ptr(xyz), n, float(voxel_size), _ptr(nv), _ptr(err), _ptr(ws), ws.numel(), st), 'spg_prune_voxels')
    V = int(nv.item())
    if int(err.item()) & 1:
        raise ValueError('prune: more than 2^21 voxels along an axis (voxel_size too small for the extent of the cloud)')
    out_xyz = torch.empty(V, 3, dtype=torch.float32, device=dev)
    out_rgb = torch.empty(V, 3, dtype=torch.uint8, device=dev)
    out_lab = torch.empty(V, n_labels + 1, dtype=torch.int32, device=dev)
    out_obj = torch.empty(V, n_objects + 1, dtype=torch.int32, device=dev)
    check(L.spg_prune_reduce(_ptr(xyz), _ptr(rgb), _ptr(labels), _ptr(objects), n, V, int(n_labels), int(n_objects), _ptr(out_xyz),
                             _ptr(out_rgb), _ptr(out_lab), _ptr(out_obj), _ptr(err), _ptr(ws), ws.numel(), st), 'spg_prune_reduce')
    if int(err.item()) & 2:
        raise IndexError('prune: a label / object id exceeds n_labels / n_objects')
    return out_xyz, out_rgb, out_lab, out_obj
