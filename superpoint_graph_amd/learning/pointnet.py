"""PointNet superpoint embedder with the reference's module API (learning/pointnet.py:16-180).

The classes own the SAME parameters under the SAME state_dict keys as the reference (they are built from
the same nn.Conv1d / nn.BatchNorm1d / nn.Linear containers in the same order, so initialisation under a
given seed is identical and reference checkpoints load), but `forward` hands the parameter pointers to
libspg_hip (spg_pointnet_forward / spg_pointnet_backward): fused 1x1-conv + BatchNorm + ReLU + max-pool
MFMA kernels instead of cuDNN/cuBLAS calls."""
import torch
import torch.nn as nn

from .. import ops


def _seq_groups(seq):
    """(weight, bias, bn.weight, bn.bias, running_mean, running_var) per parametric layer of a
    Sequential[Conv1d|Linear, (BatchNorm1d), ReLU, ...]."""
    groups, mods = [], list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            groups.append((m, bn))
        elif not isinstance(m, (nn.BatchNorm1d, nn.ReLU, nn.Dropout)):
            raise NotImplementedError(f'{type(m).__name__} is not supported by the HIP PointNet (norm must be "batch")')
        i += 1
    return groups


def _bn_momentum(bn):
    """BatchNorm1d(momentum=None) means a cumulative moving average and track_running_stats=False means batch statistics
    in eval mode: neither is implemented by the kernels (exponential average only) -- refuse instead of silently using 0.1."""
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm1d(momentum=None) (cumulative average) is not implemented on the HIP path')
    if not bn.track_running_stats or bn.running_mean is None:
        raise NotImplementedError('BatchNorm1d(track_running_stats=False) is not implemented on the HIP path')
    return bn.momentum


def _tensors(lin, bn):
    w = lin.weight
    return (w, lin.bias, None if bn is None else bn.weight, None if bn is None else bn.bias,
            None if bn is None else bn.running_mean, None if bn is None else bn.running_var)


class STNkD(nn.Module):
    """Spatial Transformer Net producing a KxK transformation matrix (reference learning/pointnet.py:16-61).
    Inside PointNet it is evaluated by the fused HIP pipeline; called on its own it runs the same kernels
    through a PointNet-less plan (K must be 2)."""

    def __init__(self, nfeat, nf_conv, nf_fc, K=2, norm='batch', affine=True, n_group=1):
        super(STNkD, self).__init__()
        if norm != 'batch':
            raise NotImplementedError('only norm="batch" is implemented on the HIP path')
        modules = []
        for i in range(len(nf_conv)):
            modules.append(nn.Conv1d(nf_conv[i - 1] if i > 0 else nfeat, nf_conv[i], 1))
            modules.append(nn.BatchNorm1d(nf_conv[i]))
            modules.append(nn.ReLU(True))
        self.convs = nn.Sequential(*modules)
        modules = []
        for i in range(len(nf_fc)):
            modules.append(nn.Linear(nf_fc[i - 1] if i > 0 else nf_conv[-1], nf_fc[i]))
            modules.append(nn.BatchNorm1d(nf_fc[i]))
            modules.append(nn.ReLU(True))
        self.fcs = nn.Sequential(*modules)
        self.proj = nn.Linear(nf_fc[-1], K * K)
        nn.init.constant_(self.proj.weight, 0)
        nn.init.constant_(self.proj.bias, 0)
        self.eye = torch.eye(K).unsqueeze(0)
        self._nfeat, self._nf_conv, self._nf_fc, self._K = nfeat, list(nf_conv), list(nf_fc), K

    def layer_groups(self):
        return _seq_groups(self.convs) + _seq_groups(self.fcs) + [(self.proj, None)]

    # ---- stand-alone evaluation (inside PointNet the STN is part of the fused pipeline) ----
    # An STN is structurally a PointNet segment without its own STN and without global features:
    # convs -> max-pool -> fcs -> plain last layer (proj); the same C entry points run it.
    def _groups_tensors(self):
        return [_tensors(lin, bn) for lin, bn in self.layer_groups()]

    def _flat_params(self):
        flat = []
        for lin, bn in self.layer_groups():
            flat += [lin.weight, lin.bias] + ([bn.weight, bn.bias] if bn is not None else [])
        return [p for p in flat if p is not None]

    def _cfg(self, npts):
        bn0 = self.convs[1]
        return ops.make_pointnet_cfg(self._nfeat, 0, 0, npts, [], [], self._nf_conv, self._nf_fc + [self._K * self._K], False,
                                     bn0.eps, _bn_momentum(bn0))

    def forward(self, input):
        """[B, nfeat, P] -> [B, K, K] transformation matrices (reference learning/pointnet.py:55-61)."""
        if not input.is_cuda:
            raise RuntimeError('superpoint_graph_amd.STNkD has no CPU path; move the module and inputs to the GPU')
        self.eye = self.eye.to(input.device)
        if self.training:
            nbt = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm1d)]
            torch._foreach_add_(nbt, 1)
        out = _PointNetFunction.apply(self, input.contiguous().float(), None, self.training, 1, *self._flat_params())
        return out.view(-1, self.eye.size(1), self.eye.size(2)) + self.eye


_anchors = {}


def _grad_anchor(device):
    """A 1-element tensor that requires grad: the single differentiable input of the HIP autograd nodes in
    FlatParameters mode."""
    a = _anchors.get(device)
    if a is None:
        a = torch.zeros(1, device=device, requires_grad=True)
        _anchors[device] = a
    return a


def _direct_grad_targets(module, groups, nparam):
    """When the module was wrapped by superpoint_graph_amd.flat.FlatParameters every parameter owns a .grad view
    into one flat, pre-zeroed gradient buffer: return those views (structure of `groups`; `None` for the bias in
    front of a BatchNorm, whose gradient is exactly zero) so that the kernels write into them directly."""
    if not getattr(module, '_spg_direct_grads', False):
        return None
    out = []
    for g in groups:
        has_bn = nparam == 4 and g[2] is not None
        row = []
        for k in range(nparam):
            t = g[k]
            if t is None or (has_bn and k == 1):
                row.append(None)
                continue
            if t.grad is None or not t.grad.is_contiguous():
                return None
            row.append(t.grad)
        out.append(tuple(row))
    return out


class _PointNetFunction(torch.autograd.Function):
    """autograd node around spg_pointnet_forward / spg_pointnet_backward; parameters are passed as inputs
    so that autograd delivers their gradients."""

    @staticmethod
    def forward(ctx, module, clouds, clouds_global, training, bn_update_times, *flat_params):
        groups = module._groups_tensors()
        emb, state = ops.pointnet_forward(module._cfg(clouds.shape[2]), clouds, clouds_global, groups, training, bn_update_times)
        ctx.module, ctx.state, ctx.groups, ctx.nflat = module, state, groups, len(flat_params)
        return emb

    @staticmethod
    def backward(ctx, grad_emb):
        direct = _direct_grad_targets(ctx.module, ctx.groups, 4)
        if direct is not None:      # gradients are written straight into the pre-assigned .grad views (FlatParameters)
            from ..flat import mark_direct_write
            mark_direct_write(ctx.module)
            ops.pointnet_backward(ctx.state, ctx.groups, grad_emb, direct)
            return (None,) * (5 + ctx.nflat)
        if ctx.nflat == 1 and getattr(ctx.module, '_spg_direct_grads', False):
            raise RuntimeError('FlatParameters mode: a parameter has no contiguous .grad view into the gradient arena '
                               '(optimizer.zero_grad(set_to_none=True) or a frozen parameter?); use FlatParameters.zero_grad()')
        gg = ops.pointnet_backward(ctx.state, ctx.groups, grad_emb)
        flat = []
        for (gw, gb, ggam, gbet) in gg:
            flat += [gw, gb, ggam, gbet]
        flat = [g for g in flat if g is not None]
        return (None, None, None, None, None) + tuple(flat)


class _LocalPointNetFunction(torch.autograd.Function):
    """PointNet WITHOUT inner STN whose xy channels are transformed by externally computed 2x2 matrices, differentiable
    wrt those matrices and wrt the global features (what LocalCloudEmbedder.run_batch composes, learning/pointnet.py:188-205)."""

    @staticmethod
    def forward(ctx, module, clouds, clouds_global, T, training, *flat_params):
        groups = module._groups_tensors()
        eye = torch.eye(2, device=T.device, dtype=T.dtype).reshape(1, 4)
        emb, state = ops.pointnet_forward(module._cfg(clouds.shape[2]), clouds, clouds_global, groups, training, 1,
                                          ext_transform=T.reshape(-1, 4) - eye)
        ctx.module, ctx.state, ctx.groups, ctx.nflat = module, state, groups, len(flat_params)
        return emb

    @staticmethod
    def backward(ctx, grad_emb):
        direct = _direct_grad_targets(ctx.module, ctx.groups, 4)
        if direct is not None:
            from ..flat import mark_direct_write
            mark_direct_write(ctx.module)
        gg, g_T, g_glob = ops.pointnet_backward(ctx.state, ctx.groups, grad_emb, direct, want_input_grads=True)
        g_T = g_T.view(-1, 2, 2)
        if direct is not None:
            return (None, None, g_glob, g_T, None) + (None,) * ctx.nflat
        flat = [g for row in gg for g in row if g is not None]
        return (None, None, g_glob, g_T, None) + tuple(flat)


class PointNet(nn.Module):
    """PointNet with one spatial transformer and a "global" input concatenated after the max-pool
    (reference learning/pointnet.py:63-133; same constructor signature)."""

    def __init__(self, nf_conv, nf_fc, nf_conv_stn, nf_fc_stn, nfeat, nfeat_stn=2, nfeat_global=1, prelast_do=0.5,
                 last_ac=False, is_res=False, norm='batch', affine=True, n_group=1, last_bn=False):
        super(PointNet, self).__init__()
        if norm != 'batch':
            raise NotImplementedError('only norm="batch" is implemented on the HIP path')
        torch.manual_seed(0)          # reference learning/pointnet.py:78
        if nfeat_stn > 0:
            self.stn = STNkD(nfeat_stn, nf_conv_stn, nf_fc_stn, norm=norm, n_group=n_group)
        self.nfeat_stn = nfeat_stn
        modules = []
        for i in range(len(nf_conv)):
            modules.append(nn.Conv1d(nf_conv[i - 1] if i > 0 else nfeat, nf_conv[i], 1))
            modules.append(nn.BatchNorm1d(nf_conv[i]))
            modules.append(nn.ReLU(True))
        self.convs = nn.Sequential(*modules)
        modules = []
        for i in range(len(nf_fc)):
            modules.append(nn.Linear(nf_fc[i - 1] if i > 0 else nf_conv[-1] + nfeat_global, nf_fc[i]))
            if i < len(nf_fc) - 1 or last_ac:
                modules.append(nn.BatchNorm1d(nf_fc[i]))
                modules.append(nn.ReLU(True))
            if i == len(nf_fc) - 2 and prelast_do > 0:
                modules.append(nn.Dropout(prelast_do))
        if is_res:
            nn.init.normal_(modules[-1].weight, mean=0, std=1e-2)
            nn.init.normal_(modules[-1].bias, mean=0, std=1e-2)
        self.fcs = nn.Sequential(*modules)
        self._nfeat, self._nfeat_global, self._last_ac, self._prelast_do = nfeat, nfeat_global, last_ac, prelast_do
        self._nf_conv, self._nf_fc = list(nf_conv), list(nf_fc)

    # ---- parameter plumbing ----
    def layer_groups(self):
        lg = self.__dict__.get('_lg_cache')
        if lg is None:                # the module structure is fixed after construction
            g = self.stn.layer_groups() if self.nfeat_stn > 0 else []
            lg = g + _seq_groups(self.convs) + _seq_groups(self.fcs)
            self.__dict__['_lg_cache'] = lg
        return lg

    def _groups_tensors(self):
        return [_tensors(lin, bn) for lin, bn in self.layer_groups()]

    def _flat_params(self):
        flat = []
        for lin, bn in self.layer_groups():
            flat += [lin.weight, lin.bias] + ([bn.weight, bn.bias] if bn is not None else [])
        return [p for p in flat if p is not None]

    def _cfg(self, npts):
        cache = self.__dict__.setdefault('_cfg_cache', {})
        if npts not in cache:
            stn_conv = self.stn._nf_conv if self.nfeat_stn > 0 else []
            stn_fc = self.stn._nf_fc if self.nfeat_stn > 0 else []
            bn0 = self.convs[1]
            cache[npts] = ops.make_pointnet_cfg(self._nfeat, self.nfeat_stn, self._nfeat_global, npts, stn_conv, stn_fc,
                                                self._nf_conv, self._nf_fc, self._last_ac, bn0.eps, _bn_momentum(bn0))
        return cache[npts]

    def _bump_batches_tracked(self, times):
        nbt = self.__dict__.get('_nbt_cache')
        if nbt is None:
            nbt = [m for m in self.modules() if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None]
            self.__dict__['_nbt_cache'] = nbt
        if nbt:
            torch._foreach_add_([m.num_batches_tracked for m in nbt], times)   # one launch for all BatchNorm layers

    def forward(self, input, input_global, bn_update_times=1):
        if not input.is_cuda:
            raise RuntimeError('superpoint_graph_amd.PointNet has no CPU path; move the model and inputs to the GPU')
        if self.training and self._prelast_do > 0:
            raise NotImplementedError('ptn_prelast_do > 0 (dropout before the last layer) is not implemented on the HIP path')
        if self.nfeat_stn > 0 and self.stn._K != 2:
            raise NotImplementedError('the fused STN applies a 2x2 transform (K=2), as PointNet.forward does')
        input = input.contiguous().float()
        if self.training:
            self._bump_batches_tracked(bn_update_times)
        if getattr(self, '_spg_direct_grads', False) and torch.is_grad_enabled():
            # FlatParameters mode: gradients are written into the arena by the kernels, so autograd only needs ONE
            # differentiable input to create the node (instead of tracking ~60 parameter tensors)
            return _PointNetFunction.apply(self, input, input_global, self.training, bn_update_times, _grad_anchor(input.device))
        return _PointNetFunction.apply(self, input, input_global, self.training, bn_update_times, *self._flat_params())


class LocalCloudEmbedder():
    """Local PointNet of the supervised partition (reference learning/pointnet.py:182-218, used by
    supervized_partition/supervized_partition.py:184,218): millions of k-nearest-neighbour clouds [n, nfeat, k] through a
    stand-alone STN (`model.stn`) and a PointNet built with nfeat_stn = 0 (`model.ptn`), embeddings L2-normalised.  The
    reference chunks the batch for cuDNN (2^16 - 1 clouds per call, :193) and materialises the transformed clouds; here the
    2x2 transform is applied by the first convolution while it stages the cloud.  The kernels have no chunk limit, but the
    reference's chunks are OBSERVABLE in training mode (every chunk is its own BatchNorm batch: own statistics, own
    running-stat update), so training-mode batches above 2^16 - 1 clouds are processed in the same chunks; in eval mode
    (running statistics) the whole batch is one launch sequence."""

    CHUNK = 2 ** 16 - 1      # learning/pointnet.py:193

    def __init__(self, args):
        self.nfeat_stn = args.ptn_nfeat_stn
        self.stn_as_global = args.stn_as_global

    def run_batch(self, model, clouds, clouds_global, *excess):
        if not clouds.is_cuda:
            raise RuntimeError('superpoint_graph_amd.LocalCloudEmbedder has no CPU path')
        n = clouds.shape[0]
        if n > self.CHUNK and (model.ptn.training or (self.nfeat_stn > 0 and model.stn.training)):
            # the reference's order: all STN chunks, then all PointNet chunks (:196-198, :204-206); the two networks share no
            # BatchNorm layer, so chunk-by-chunk evaluation of the pair gives the same statistics and updates
            clouds_global = clouds_global.reshape(n, -1)
            return torch.cat([self._run(model, clouds[a:a + self.CHUNK], clouds_global[a:a + self.CHUNK])
                              for a in range(0, n, self.CHUNK)])
        return self._run(model, clouds, clouds_global)

    def _run(self, model, clouds, clouds_global):
        ptn = model.ptn
        if ptn.nfeat_stn > 0:
            raise ValueError('LocalCloudEmbedder expects model.ptn without an inner STN (nfeat_stn = 0) and a separate model.stn')
        clouds = clouds.contiguous().float()
        clouds_global = clouds_global.float().reshape(clouds.shape[0], -1)
        if self.nfeat_stn > 0:
            T = model.stn(clouds[:, :self.nfeat_stn, :].contiguous())
            if self.stn_as_global:
                clouds_global = torch.cat([clouds_global, T.reshape(-1, 4)], 1)
            if ptn.training:
                ptn._bump_batches_tracked(1)
            extra = (_grad_anchor(clouds.device),) if getattr(ptn, '_spg_direct_grads', False) and torch.is_grad_enabled() else tuple(ptn._flat_params())
            out = _LocalPointNetFunction.apply(ptn, clouds, clouds_global.contiguous(), T, ptn.training, *extra)
        else:
            out = ptn(clouds, clouds_global.contiguous())
        return nn.functional.normalize(out)

    def run_batch_cpu(self, model, clouds, clouds_global, *excess):
        """Embeddings on the host (the reference moves 1023-cloud chunks to the CPU as they are computed to bound GPU
        memory, :207-218); with 288 GB of HBM the whole batch is embedded at once and copied back."""
        return self.run_batch(model, clouds, clouds_global).cpu()


class _ScatterEmbeddings(torch.autograd.Function):
    """descriptors[i] = embeddings[slot[i]] for embeddable superpoints, exact zeros for the others (reference
    learning/pointnet.py:177-179: zero-initialised matrix + index_copy_), one launch each way instead of fill + index_copy_."""

    @staticmethod
    def forward(ctx, out, slot_of_row, idx_valid):
        ctx.idx_valid = idx_valid
        return ops.gather_rows(out.contiguous(), slot_of_row)

    @staticmethod
    def backward(ctx, grad):
        return ops.gather_rows(grad.contiguous(), ctx.idx_valid), None, None


def flag_index_vectors(clouds_flag):
    """HOST index vectors CloudEmbedder derives from `clouds_flag` (learning/pointnet.py:149,162,177-179): rows of the valid
    superpoints, and the embedding row of every superpoint (-1 for the too-small ones) -> (idx_valid, slot_of_row), int64."""
    valid = clouds_flag.eq(0)
    idx_valid = torch.nonzero(valid).reshape(-1)
    slot = torch.cumsum(valid.to(torch.int64), 0) - 1
    slot[~valid] = -1
    return idx_valid, slot


def attach_staged_flags(clouds_flag, idx_valid_dev, slot_dev):
    """The device copies of flag_index_vectors(clouds_flag), uploaded by the caller (a device collate packs them into the batch's one
    staging copy: GraphConvInfo.set_batch_device(extras=...)), attached to the flag tensor for CloudEmbedder."""
    clouds_flag._spg_staged = (idx_valid_dev, slot_dev)
    return clouds_flag._spg_staged


def stage_flags(clouds_flag):
    """Index vectors CloudEmbedder derives from `clouds_flag` (rows of the valid superpoints, embedding row of every
    superpoint), computed and uploaded on the CURRENT stream and attached to the flag tensor: a device collate calls this
    while it builds the batch (on the side stream of learning/prefetch.py), so the training stream starts a step without the
    two small uploads.  -> (idx_valid, slot_of_row) on the device."""
    staged = getattr(clouds_flag, '_spg_staged', None)
    if staged is None:
        dev = torch.device('cuda', torch.cuda.current_device())
        idx_valid, slot = flag_index_vectors(clouds_flag)
        # (staging ring, non-blocking: a pageable H2D would stall the host until the stream has drained)
        if clouds_flag.is_cuda:       # (a flag vector that lives on the device already: its index vectors do too)
            staged = clouds_flag._spg_staged = (idx_valid, slot)
        else:
            staged = clouds_flag._spg_staged = tuple(ops.upload_packed([idx_valid, slot], dev))
    return staged


class CloudEmbedder():
    """Evaluates PointNet on superpoints; too small superpoints get zero embeddings (reference
    learning/pointnet.py:138-180).  `ptn_mem_monger` keeps its observable semantics (autograd is cut after
    the embeddings, `bw_hook()` back-propagates into PointNet, BatchNorm running statistics advance twice
    per step) but nothing is recomputed: the raw layer outputs stay in HBM."""

    def __init__(self, args):
        self.args = args
        self.bw_hook = lambda: None
        self.run = self.run_full_monger if args.ptn_mem_monger else self.run_full
        self._flag_cache = (None, None, None)      # (clouds_flag tensor, idx_valid, slot_of_row on the device)

    def _to_device(self, clouds_flag, clouds, clouds_global):
        dev = torch.device('cuda', torch.cuda.current_device())
        if self._flag_cache[0] is not clouds_flag:       # (same batch object again: benchmarks, multi-pass evaluation)
            self._flag_cache = (clouds_flag,) + tuple(stage_flags(clouds_flag))      # staged by the device collate, or now
        self._slot_of_row = self._flag_cache[2]
        return self._flag_cache[1], ops.upload(clouds, dev), ops.upload(clouds_global, dev)

    def _scatter(self, out, idx_valid, n_rows):
        if out.shape[0] == 0:                          # no embeddable superpoint in the batch
            return out.new_zeros(n_rows, out.size(1))
        return _ScatterEmbeddings.apply(out, self._slot_of_row, idx_valid)

    def run_full(self, model, clouds_meta, clouds_flag, clouds, clouds_global):
        if not self.args.cuda:
            raise RuntimeError('superpoint_graph_amd has no CPU path (--cuda 1 required)')
        idx_valid, clouds, clouds_global = self._to_device(clouds_flag, clouds, clouds_global)
        out = model.ptn(clouds, clouds_global)
        return self._scatter(out, idx_valid, clouds_flag.size(0))

    def run_full_monger(self, model, clouds_meta, clouds_flag, clouds, clouds_global):
        if not self.args.cuda:
            raise RuntimeError('superpoint_graph_amd has no CPU path (--cuda 1 required)')
        idx_valid, clouds, clouds_global = self._to_device(clouds_flag, clouds, clouds_global)
        ptn = model.ptn
        if not model.training:
            with torch.no_grad():
                out = ptn(clouds, clouds_global)
            self.bw_hook = lambda: None
        else:
            # one forward that keeps its activations; the running statistics advance as in the reference's
            # forward + re-forward (learning/pointnet.py:167,173)
            live = ptn(clouds, clouds_global, bn_update_times=2)
            out = live.detach().requires_grad_(True)

            def bw_hook():
                if out.grad is not None:
                    live.backward(out.grad)
            self.bw_hook = bw_hook
        return self._scatter(out, idx_valid, clouds_flag.size(0))
