"""FusedStep: forward + backward of one training step as ONE call into libspg_hip (include/spg_hip.h: spg_train_step).

The reference's loop body (learning/main.py:199-208)
    embeddings = ptnCloudEmbedder.run(model, *clouds_data); outputs = model.ecc(embeddings)
    loss = cross_entropy(outputs, label_mode, weight=class_weights); loss.backward(); ptnCloudEmbedder.bw_hook()
costs ~15 autograd nodes / ctypes calls and, on the device, orders the filter network's forward in front of the recurrence
and the tail of the RNN-ECC backward in front of PointNet's backward although neither depends on PointNet.  For the standard
model -- `gru_R...` / `lstm_R...` followed by `f_K` (every documented configuration) -- with its parameters in a
FlatParameters arena this object issues the same kernels through one C call that knows the whole step (the two chains travel
next to PointNet's launches, superpoint_graph_amd/csrc/spg_step.hip).  For GRU models the classifier and the cross entropy
run inside the one-launch recurrence (per node, in the wavefront that owns it).  Results are those of the
module-level path: tests/test_gpu_fused.py compares loss, logits, every gradient and the BatchNorm statistics bit for bit with
the classifier / loss as separate launches (spg_tune key 15 = 1) and at fp32 round-off in the default form.

What it keeps of the module API's observable behaviour: BatchNorm batch counters advance (twice for PointNet with
`ptn_mem_monger`, like the reference's forward + re-forward), too-small superpoints get exact-zero descriptors, the gradients
land in `p.grad` (the arena), `arena.zero_grad()` / `optimizer_step()` work as before.  What it does not do: autograd -- the
returned loss / logits are detached; anything outside `supports(model)` must use the modules."""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from .flat import mark_direct_write


def supports(model) -> bool:
    """True for `model.ptn` = PointNet (inner STN or none) and `model.ecc` = GraphNetwork of exactly one recurrent graph
    convolution followed by one linear layer that the HIP kernels serve."""
    from .learning.modules import HipLinear, RNNGraphConvModule
    from .learning.pointnet import PointNet
    ptn, ecc = getattr(model, 'ptn', None), getattr(model, 'ecc', None)
    if not isinstance(ptn, PointNet) or ecc is None:
        return False
    kids = list(ecc.children())
    if len(kids) != 2 or not isinstance(kids[0], RNNGraphConvModule) or not isinstance(kids[1], HipLinear):
        return False
    conv, fc = kids
    if conv._cell.hidden_size != 32 or conv._cell.input_size != 32 or conv._cell.bias_ih is None:
        return False
    return fc._kernel_shape_ok() and ptn._prelast_do == 0 and (ptn.nfeat_stn == 0 or ptn.stn._K == 2)


class FusedStep:
    def __init__(self, model, arena, class_weights=None, reduction='mean', ignore_index=-100, ptn_mem_monger=True):
        if not supports(model):
            raise NotImplementedError('FusedStep serves PointNet + GraphNetwork("gru_R.../lstm_R...,f_K") only')
        if arena is None or arena.model is not model:
            raise ValueError('FusedStep needs the FlatParameters arena of this model (the kernels write the gradients in place)')
        if reduction not in ('mean', 'sum'):
            raise NotImplementedError("reduction must be 'mean' or 'sum'")
        self.model, self.arena = model, arena
        self.ptn = model.ptn
        self.conv, self.fc = list(model.ecc.children())
        # (validated like every other device operand: a CPU / other-GPU tensor here would reach the kernels as a wild pointer)
        self.class_weights = None if class_weights is None else ops._req(class_weights.detach().float().contiguous(), torch.float32, 'class_weights')
        if self.class_weights is not None and self.class_weights.numel() != list(model.ecc.children())[1].out_features:
            raise ValueError('class_weights: one weight per class expected')
        self.mean = reduction == 'mean'
        self.ignore_index = int(ignore_index)
        self.bn_times = 2 if ptn_mem_monger else 1
        self._plan = None
        self._bufs = {}

    # ---- parameter / gradient pointer tables: the arena's views never move ----
    def _tables(self, npts):
        if self._plan is not None and self._plan['npts'] == npts:
            return self._plan
        ptn, conv, fc = self.ptn, self.conv, self.fc
        ptn_cfg = ptn._cfg(npts)
        ecc_cfg, ecc_groups = conv._cfg_and_groups()
        pg = ptn._groups_tensors()

        def grads_of(groups, nparam):
            rows = []
            for g in groups:
                has_bn = nparam == 4 and g[2] is not None
                row = []
                for k in range(6):
                    t = g[k] if k < nparam else None
                    if t is None or (has_bn and k == 1):      # a bias in front of train-mode BatchNorm: zero gradient, never written
                        row.append(None)
                    else:
                        if t.grad is None or not t.grad.is_contiguous():
                            raise RuntimeError('FusedStep: a parameter has no contiguous .grad view into the gradient arena')
                        row.append(t.grad)
                rows += row
            return rows
        nf = ecc_cfg.n_fnet
        keep = [t for g in pg for t in g] + [t for g in ecc_groups for t in g]
        ptn_grads = grads_of(pg, 4)
        ecc_grads = grads_of(ecc_groups[:nf], 4) + grads_of(ecc_groups[nf:], 6)
        keep += ptn_grads + ecc_grads
        self._plan = dict(npts=npts, ptn_cfg=ptn_cfg, ecc_cfg=ecc_cfg,
                          ptn_params=ops._ptr_array([t for g in pg for t in g]), ptn_grads=ops._ptr_array(ptn_grads),
                          ecc_params=ops._ptr_array([t for g in ecc_groups for t in g]), ecc_grads=ops._ptr_array(ecc_grads),
                          keep=keep, nf=int(ptn_cfg.fc[ptn_cfg.n_fc - 1]),
                          nout=int(ecc_cfg.nc * (ecc_cfg.nrepeats + 1) if ecc_cfg.cat_all else ecc_cfg.nc))
        return self._plan

    def _buffers(self, plan, B, N, E, dev, ecc_cfg):
        """Workspaces and activations of one step, re-used from step to step while the sizes stay the same (one stream: the
        next step overwrites them only after this one has consumed them)."""
        # (npts: the PointNet plan -- hence the workspace layout and the statistics slots -- depends on the points per cloud)
        key = (B, N, E, dev, plan['npts'], tuple(ecc_cfg.part_ptr[:ecc_cfg.n_parts + 1]) if ecc_cfg.n_parts > 0 else ())
        b = self._bufs.get(key)
        if b is not None:
            return b
        L = _lib.lib()
        pc, ec = ctypes.byref(plan['ptn_cfg']), ctypes.byref(ecc_cfg)
        nf, nout, C = plan['nf'], plan['nout'], self.fc.out_features
        u8 = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        sizes = (L.spg_pointnet_workspace_bytes(pc, B, 1), L.spg_pointnet_bwd_workspace_bytes(pc, B),
                 L.spg_eccrnn_workspace_bytes(ec, N, E, 1), L.spg_eccrnn_bwd_workspace_bytes(ec, N, E))
        if min(sizes) == 0:
            raise RuntimeError('workspace query failed: ' + L.spg_last_error().decode())
        b = dict(ptn_ws=u8(sizes[0]), ptn_bwd_ws=u8(sizes[1]), ecc_ws=u8(sizes[2]), ecc_bwd_ws=u8(sizes[3]),
                 emb=f32(B, nf), grad_emb=f32(B, nf), desc=f32(N, nf), grad_desc=f32(N, nf), ecc_out=f32(N, nout), grad_ecc_out=f32(N, nout),
                 cls_work=f32(max(1, L.spg_linear_wgrad_bias_work_floats(N, C, nout))), grad_logits=f32(N, C))
        if len(self._bufs) >= 4:       # a few batch shapes at most stay cached (training batches vary in size)
            self._bufs.pop(next(iter(self._bufs)))
        self._bufs[key] = b
        return b

    def __call__(self, clouds_flag, clouds, clouds_global, gc_info, target):
        """-> (loss [] float32, logits [N, n_classes]) on the device, detached; gradients in the arena.
        clouds_flag / clouds / clouds_global: as CloudEmbedder.run takes them; gc_info: the batch's GraphConvInfo (already on
        the device: GraphNetwork.set_info(..., cuda=True) or set_batch_device); target: int64 [N] device tensor."""
        from .learning.pointnet import stage_flags
        model, ptn, conv, fc = self.model, self.ptn, self.conv, self.fc
        if not (ptn.training and conv.training):
            raise RuntimeError('FusedStep is a TRAINING step (model.train()); evaluate through the modules')
        dev = torch.device('cuda', torch.cuda.current_device())
        idx_valid, slot_of_row = stage_flags(clouds_flag)
        clouds = ops.upload(clouds, dev) if not clouds.is_cuda else clouds
        clouds_global = ops.upload(clouds_global, dev) if not clouds_global.is_cuda else clouds_global
        clouds = ops._req(clouds.contiguous(), torch.float32, 'clouds')
        B, N = int(clouds.shape[0]), int(clouds_flag.shape[0])
        if B <= 1:
            raise ValueError(f'Expected more than 1 value per channel when training, got input size [{B}, C]')
        plan = self._tables(int(clouds.shape[2]))
        if clouds.shape[1] != plan['ptn_cfg'].nfeat:
            raise ValueError('clouds: wrong number of point features')
        clouds_global = ops._req(clouds_global.reshape(B, -1).contiguous(), torch.float32, 'clouds_global')
        idxn, idxe, degs, degs_gpu, edgefeats = gc_info.get_buffers()
        if idxe is not None:
            raise NotImplementedError('filter sharing (idxe) is not supported by the fused RNN-ECC path')
        graph = gc_info.device_graph()
        E = int(graph.E)
        if graph.N != N:
            raise ValueError(f'the batched graph has {graph.N} nodes, clouds_flag has {N} rows')
        edgefeats = ops._req(edgefeats.contiguous().float(), torch.float32, 'edgefeats')
        if plan['ecc_cfg'].bnidx >= 0 and E == 1:
            raise ValueError('Expected more than 1 value per channel when training (filter-network BatchNorm over one edge)')
        target = ops._req(target.contiguous(), torch.int64, 'target')
        ecc_cfg, _ = conv._cfg_for(gc_info, N)              # (+ the batch's scene boundaries for graphs above one round)
        b = self._buffers(plan, B, N, E, dev, ecc_cfg)
        C = fc.out_features
        logits = torch.empty(N, C, dtype=torch.float32, device=dev)
        loss_buf = torch.empty(N + 2, dtype=torch.float32, device=dev)
        a = _lib.StepArgs()
        a.ptn_cfg, a.B, a.bn_update_times = ctypes.pointer(plan['ptn_cfg']), B, self.bn_times
        a.clouds, a.clouds_global = clouds.data_ptr(), clouds_global.data_ptr()
        a.ptn_params, a.ptn_grads = plan['ptn_params'], plan['ptn_grads']
        a.ptn_ws, a.ptn_bwd_ws, a.emb, a.grad_emb = b['ptn_ws'].data_ptr(), b['ptn_bwd_ws'].data_ptr(), b['emb'].data_ptr(), b['grad_emb'].data_ptr()
        a.N, a.nf = N, plan['nf']
        a.slot_of_row, a.idx_valid = slot_of_row.data_ptr(), idx_valid.data_ptr()
        a.desc, a.grad_desc = b['desc'].data_ptr(), b['grad_desc'].data_ptr()
        a.ecc_cfg, a.E, a.graph_ws = ctypes.pointer(ecc_cfg), E, graph.ws.data_ptr()
        a.edgefeats = edgefeats.data_ptr() if E else None
        a.ecc_params, a.ecc_grads = plan['ecc_params'], plan['ecc_grads']
        a.ecc_ws, a.ecc_bwd_ws = b['ecc_ws'].data_ptr(), b['ecc_bwd_ws'].data_ptr()
        a.ecc_out, a.grad_ecc_out = b['ecc_out'].data_ptr(), b['grad_ecc_out'].data_ptr()
        a.nout, a.n_classes = plan['nout'], C
        a.cls_W, a.cls_b = fc.weight.data_ptr(), None if fc.bias is None else fc.bias.data_ptr()
        if fc.weight.grad is None or not fc.weight.grad.is_contiguous() or (fc.bias is not None and fc.bias.grad is None):
            raise RuntimeError('FusedStep: the classifier has no contiguous .grad view into the gradient arena')
        a.cls_dW, a.cls_db = fc.weight.grad.data_ptr(), None if fc.bias is None else fc.bias.grad.data_ptr()
        a.cls_work, a.logits, a.grad_logits = b['cls_work'].data_ptr(), logits.data_ptr(), b['grad_logits'].data_ptr()
        a.target = target.data_ptr()
        a.class_weight = None if self.class_weights is None else self.class_weights.data_ptr()
        a.ignore_index, a.reduction_mean, a.loss_buf = self.ignore_index, int(self.mean), loss_buf.data_ptr()
        # the statistics slots inside ptn_ws are zero after every successful step on these buffers (include/spg_hip.h)
        a.ptn_slots_clean = 1 if b.get('slots_clean') else 0
        b['slots_clean'] = False
        _lib.check(_lib.lib().spg_train_step(ctypes.byref(a), ops._stream()), 'spg_train_step')
        b['slots_clean'] = True
        # module-level bookkeeping the C call does not do -- only once the call has succeeded (a refused step must not advance
        # the BatchNorm batch counters or mark gradients as written)
        ptn._bump_batches_tracked(self.bn_times)
        for m in conv._fnet:
            if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None:
                m.num_batches_tracked += 1
        for m in (ptn, conv, fc):
            mark_direct_write(m)
        self._last = (plan['ptn_cfg'], B, b['ptn_ws'], ecc_cfg, graph, b['ecc_ws'], edgefeats)
        self.normaliser = loss_buf[N + 1:N + 2]       # sum of the labelled rows' class weights (data-parallel loss weight w_r)
        self._emb, self._slot = b['emb'], slot_of_row   # (the descriptors are read in place by the recurrence: see `embeddings`)
        return loss_buf[N], logits

    def infer(self, clouds_flag, clouds, clouds_global, gc_info):
        """-> logits [N, n_classes] of the model in EVALUATION mode (model.eval(): BatchNorm running statistics), the body of
        the evaluation loop (learning/main.py:256-262) as ONE library call (spg_infer_step); same kernels and results as
        `model.ecc(CloudEmbedder.run(...))` under torch.no_grad()."""
        from .learning.pointnet import stage_flags
        ptn, conv, fc = self.ptn, self.conv, self.fc
        if ptn.training or conv.training:
            raise RuntimeError('FusedStep.infer is the EVALUATION forward (model.eval()); train with __call__')
        dev = torch.device('cuda', torch.cuda.current_device())
        idx_valid, slot_of_row = stage_flags(clouds_flag)
        clouds = ops.upload(clouds, dev) if not clouds.is_cuda else clouds
        clouds_global = ops.upload(clouds_global, dev) if not clouds_global.is_cuda else clouds_global
        clouds = ops._req(clouds.contiguous(), torch.float32, 'clouds')
        B, N = int(clouds.shape[0]), int(clouds_flag.shape[0])
        if B < 1:
            raise ValueError('no embeddable superpoint in the batch')
        plan = self._tables(int(clouds.shape[2]))
        if clouds.shape[1] != plan['ptn_cfg'].nfeat:
            raise ValueError('clouds: wrong number of point features')
        clouds_global = ops._req(clouds_global.reshape(B, -1).contiguous(), torch.float32, 'clouds_global')
        idxn, idxe, degs, degs_gpu, edgefeats = gc_info.get_buffers()
        if idxe is not None:
            raise NotImplementedError('filter sharing (idxe) is not supported by the fused RNN-ECC path')
        graph = gc_info.device_graph()
        E = int(graph.E)
        if graph.N != N:
            raise ValueError(f'the batched graph has {graph.N} nodes, clouds_flag has {N} rows')
        edgefeats = ops._req(edgefeats.contiguous().float(), torch.float32, 'edgefeats')
        ecc_cfg, _ = conv._cfg_for(gc_info, N)
        key = ('infer', B, N, E, dev)
        b = self._bufs.get(key)
        if b is None:
            L = _lib.lib()
            pc, ec = ctypes.byref(plan['ptn_cfg']), ctypes.byref(ecc_cfg)
            sizes = (L.spg_pointnet_workspace_bytes(pc, B, 0), L.spg_eccrnn_workspace_bytes(ec, N, E, 0))
            if min(sizes) == 0:
                raise RuntimeError('workspace query failed: ' + L.spg_last_error().decode())
            u8 = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
            f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            b = dict(ptn_ws=u8(sizes[0]), ecc_ws=u8(sizes[1]), emb=f32(B, plan['nf']), desc=f32(N, plan['nf']), ecc_out=f32(N, plan['nout']))
            if len(self._bufs) >= 4:
                self._bufs.pop(next(iter(self._bufs)))
            self._bufs[key] = b
        C = fc.out_features
        logits = torch.empty(N, C, dtype=torch.float32, device=dev)
        a = _lib.StepArgs()
        a.ptn_cfg, a.B, a.bn_update_times = ctypes.pointer(plan['ptn_cfg']), B, 1
        a.clouds, a.clouds_global = clouds.data_ptr(), clouds_global.data_ptr()
        a.ptn_params, a.ptn_ws, a.emb = plan['ptn_params'], b['ptn_ws'].data_ptr(), b['emb'].data_ptr()
        a.N, a.nf = N, plan['nf']
        a.slot_of_row, a.idx_valid, a.desc = slot_of_row.data_ptr(), idx_valid.data_ptr(), b['desc'].data_ptr()
        a.ecc_cfg, a.E, a.graph_ws = ctypes.pointer(ecc_cfg), E, graph.ws.data_ptr()
        a.edgefeats = edgefeats.data_ptr() if E else None
        a.ecc_params, a.ecc_ws, a.ecc_out = plan['ecc_params'], b['ecc_ws'].data_ptr(), b['ecc_out'].data_ptr()
        a.nout, a.n_classes = plan['nout'], C
        a.cls_W, a.cls_b, a.logits = fc.weight.data_ptr(), None if fc.bias is None else fc.bias.data_ptr(), logits.data_ptr()
        _lib.check(_lib.lib().spg_infer_step(ctypes.byref(a), ops._stream()), 'spg_infer_step')
        return logits

    def debug_states(self):
        """(PointNetState, EccRnnState) views of the LAST training step's forward workspaces -- what ops.pointnet_forward /
        ops.eccrnn_forward return on the module path; the parity tests read the kernels' ReLU / max-pool decisions out of them
        (spg_pointnet_debug_offset / spg_eccrnn_debug_offset).  Valid until the next step on the same buffers."""
        ptn_cfg, B, ptn_ws, ecc_cfg, graph, ecc_ws, edgefeats = self._last
        return (ops.PointNetState(ptn_cfg, B, None, None, ptn_ws, True), ops.EccRnnState(ecc_cfg, graph, edgefeats, ecc_ws, True))

    @property
    def embeddings(self):
        """[N, nf] superpoint descriptors of the last step (CloudEmbedder.run's return value: PointNet embeddings scattered to all
        superpoints, zero rows for the too-small ones) -- built on request: the step itself never materialises them when the
        one-launch recurrence reads the scatter in place."""
        return ops.gather_rows(self._emb, self._slot)
