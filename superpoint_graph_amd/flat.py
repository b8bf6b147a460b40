"""FlatParameters: every trainable parameter of a model (and its gradient) becomes a view into ONE flat fp32 buffer.

Why (MI355X-first, launch-bound regime): a training step of the 279k-parameter model is ~230 dependent kernel
launches; with 69 separate parameter tensors the reference-style update adds 69 gradient accumulations, a
multi-tensor Adam and 69 clamps on top.  With flat storage
  * the HIP backward kernels write the gradients straight into their final location (no per-parameter allocation,
    accumulation or zero-fill launches; one fill per step zeroes the whole arena),
  * the element-wise clamp (learning/main.py:210-212) and Adam are ONE launch sequence over one tensor,
  * the data-parallel all-reduce (superpoint_graph_amd/dist.py) runs on the arena itself: no flatten/unflatten.
The arithmetic is unchanged (clamp and Adam are element-wise), `state_dict()` keys / shapes are unchanged."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class FlatParameters:
    def __init__(self, model: torch.nn.Module, lazy_zero: bool = False, host_counters: bool = False):
        """lazy_zero: `zero_grad()` does not launch a fill.  The HIP backward kernels OVERWRITE every gradient view of the
        modules they serve, so zeroing the arena first is redundant work in the fixed loop zero_grad -> forward -> backward ->
        step (learning/main.py:199-213); the gradients of parameters no kernel wrote since the last zero_grad() (a module that
        did not take part in the step) are zeroed right before they are consumed (adam_step / allreduce / clamp_grad_).
        Observable difference: between zero_grad() and backward(), `p.grad` still shows the previous step's values.
        host_counters: the BatchNorm `num_batches_tracked` buffers (pure bookkeeping on this path: momentum is fixed) move to
        the host, so advancing them costs no kernel launch; state_dict() / load_state_dict() work as before."""
        self.model = model
        self.lazy_zero = bool(lazy_zero)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('model has no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        # every parameter starts on a 256-byte boundary: the kernels' 16-byte vector paths need aligned bases
        self._offsets, off = [], 0
        for p in self.params:
            self._offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        data = torch.zeros(self.numel, dtype=dt, device=dev)
        self._gbuf = torch.zeros(self.numel + 3, dtype=dt, device=dev)       # +3: loss-weight, loss-sum and time-out-flag slots of the all-reduce
        self._guard_all = None               # set by allreduce_sums: the ranks' summed time-out flags (read by the next adam_step on the device)
        for p, off in zip(self.params, self._offsets):
            n = p.numel()
            data[off:off + n].copy_(p.data.reshape(-1))
            p.data = data[off:off + n].view(p.shape)
            p.grad = self._gbuf[off:off + n].view(p.shape)
        self.flat = torch.nn.Parameter(data)             # hand THIS to the optimizer
        self.flat.grad = self._gbuf[:self.numel]
        self._modules = list(model.modules())
        self._written_log = []                           # modules whose backward has written since the last _clear_written()
        self._uncovered = {}                             # frozenset(written module ids) -> parameters no kernel has written
        for m in self._modules:
            m._spg_direct_grads = True                   # the HIP autograd Functions then write into p.grad directly
            m._spg_grad_written = False
            m._spg_written_log = self._written_log
        # for every parameter: the modules on its path (a kernel-backed module writes the gradients of all its descendants)
        named = dict(model.named_modules())
        self._cover = []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            parts = name.split('.')[:-1]
            self._cover.append([named['.'.join(parts[:k])] for k in range(len(parts) + 1)])
        # Parameters NO kernel-backed module covers (an affine `b` BatchNorm token of GraphNetwork, a HipLinear whose input
        # width is no multiple of 4 and therefore runs torch's F.linear, any stock torch layer a user adds) receive their
        # gradients from autograd's AccumulateGrad, which ADDS to `.grad`: these views are zeroed eagerly by zero_grad()
        # -- one multi-tensor fill, and only when such parameters exist -- and are never touched by _resolve_stale().
        self._autograd_grads = [p.grad for p, chain in zip(self.params, self._cover)
                                if not any(_direct_writer(m) for m in chain)]
        self._covered = [(p, chain) for p, chain in zip(self.params, self._cover) if any(_direct_writer(m) for m in chain)]
        for m in self._modules:
            m._spg_lazy_zero = self.lazy_zero
        self._stale = False
        self.one = torch.ones((), dtype=dt, device=dev)      # seed of loss.backward(arena.one): autograd otherwise launches a fill for it
        if host_counters:
            for m in self._modules:
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None:
                    m.num_batches_tracked = m.num_batches_tracked.cpu()

    def adam_step(self, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_clip=0.0, grad_div=None):
        """Element-wise gradient clamp (learning/main.py:210-212) + torch.optim.Adam update (learning/main.py:433-437)
        of the whole arena in ONE HIP launch (spg_adam_clamp_step).  The moments live in this object.
        grad_div: [1] device tensor; every gradient is divided by it first (`allreduce_sums` leaves the data-parallel
        normaliser in `self.normaliser`)."""
        from . import _lib
        if not hasattr(self, '_m'):
            self._m = torch.zeros_like(self.flat.data)
            self._v = torch.zeros_like(self.flat.data)
            self._t = 0
        self._resolve_stale()
        self._t += 1
        guard_all = self._guard_all             # (one use: the flag belongs to the exchange of THIS step)
        self._guard_all = None
        _lib.check(_lib.lib().spg_adam_clamp_step_guarded(self.flat.data.data_ptr(), self.flat.grad.data_ptr(), self._m.data_ptr(),
                                                          self._v.data_ptr(), self.numel, lr, betas[0], betas[1], eps, weight_decay,
                                                          grad_clip, self._t, None if grad_div is None else grad_div.data_ptr(),
                                                          None if guard_all is None else guard_all.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), 'spg_adam_clamp_step')
        self._clear_written()

    def rewind_steps(self, n):
        """Takes n optimiser updates that the device WITHHELD (the fail-safe of spg_adam_clamp_step: a persistent RNN-ECC launch had
        timed out -- ops.recover_persistent_ecc) off the host's step counter, so that Adam's bias correction continues from the
        update the parameters have really seen."""
        n = int(n)
        if n <= 0 or not hasattr(self, '_t'):
            return
        self._t = max(self._t - n, 0)
        if getattr(self, '_step_t', None) is not None:
            self._step_t.fill_(float(self._t))

    def attach_optimizer(self, optimizer):
        """Keeps a `torch.optim.Adam` as the owner of the hyper-parameters (learning-rate schedulers keep working) and of the
        state dict (checkpoints stay in torch's format, both ways): its per-parameter moments become views of the arena's
        flat moment buffers -- moments loaded from a checkpoint are copied in -- and `optimizer_step()` replaces
        `optimizer.step()`."""
        if not isinstance(optimizer, torch.optim.Adam) or len(optimizer.param_groups) != 1:
            raise TypeError('the fused update implements torch.optim.Adam with one parameter group')
        if optimizer.param_groups[0].get('amsgrad', False) or optimizer.param_groups[0].get('maximize', False):
            raise NotImplementedError('amsgrad / maximize are not implemented by the fused update')
        self._opt = optimizer
        self._m = torch.zeros_like(self.flat.data)
        self._v = torch.zeros_like(self.flat.data)
        self._t = 0
        self._step_t = torch.tensor(0.0)               # ONE step tensor shared by every parameter's state
        for p, off in zip(self.params, self._offsets):
            n = p.numel()
            m, v = self._m[off:off + n].view(p.shape), self._v[off:off + n].view(p.shape)
            st = optimizer.state.get(p)
            if st:                                     # resumed run
                m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
                self._t = int(st['step'])
            optimizer.state[p] = {'step': self._step_t, 'exp_avg': m, 'exp_avg_sq': v}
        self._step_t.fill_(float(self._t))

    def optimizer_step(self, grad_clip=0.0, grad_div=None):
        g = self._opt.param_groups[0]
        self.adam_step(lr=g['lr'], betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'], grad_clip=grad_clip,
                       grad_div=grad_div)
        self._step_t.add_(1.0)

    # ---- data parallel, everything on the device ----
    @property
    def normaliser(self):
        """[1] view: after `allreduce_sums` the sum over the ranks of the loss weights (pass it to adam_step(grad_div=...))."""
        return self._gbuf[self.numel:self.numel + 1]

    @property
    def loss_sum(self):
        """[1] view: after `allreduce_sums` the sum over the ranks of the (weighted, un-normalised) losses."""
        return self._gbuf[self.numel + 1:self.numel + 2]

    def allreduce_sums(self, weight, loss_sum=None, group=None):
        """The data-parallel exchange of one step as ONE collective and no host synchronisation.  Every rank back-propagated
        its SUM-reduced loss (gradients g_r already carry the factor w_r of superpoint_graph_amd/dist.py), `weight` is its [1]
        device tensor w_r (ops.cross_entropy(..., return_normaliser=True)), `loss_sum` optionally its loss.  The arena
        [gradients | w_r | loss_r] is summed over the ranks in place; the division by sum_r w_r happens inside the clamp + Adam
        launch (adam_step(grad_div=self.normaliser)).  World size 1 (or no process group): the same code without the
        collective -- the result is the single-process gradient of the mean-reduced loss."""
        from . import _lib
        self._resolve_stale()
        self._gbuf[self.numel:self.numel + 1].copy_(weight.reshape(1), non_blocking=True)
        if loss_sum is not None:
            self._gbuf[self.numel + 1:self.numel + 2].copy_(loss_sum.detach().reshape(1), non_blocking=True)
        # the fail-safe of the one-launch RNN-ECC recurrences across ranks: this rank's time-out flag (0 / 1, written on the device)
        # travels as a third slot; the collective sums the flags and the clamp + Adam launch of EVERY rank withholds its update
        # while the sum is non-zero -- the summed gradients contain the failed rank's wrong ones (include/spg_hip.h:
        # spg_adam_clamp_step_guarded).  No host synchronisation.
        if self._gbuf.is_cuda:
            flag = self._gbuf[self.numel + 2:self.numel + 3]
            _lib.check(_lib.lib().spg_ecc_persistent_flag(flag.data_ptr(), torch.cuda.current_stream().cuda_stream), 'spg_ecc_persistent_flag')
            self._guard_all = flag
        native = _lib.lib().spg_rccl_world_size()
        if native > 1:
            _lib.check(_lib.lib().spg_rccl_allreduce_sum_f32(self._gbuf.data_ptr(), self.numel + 3,
                                                             torch.cuda.current_stream().cuda_stream), 'spg_rccl_allreduce_sum_f32')
        elif dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == 'gloo' and self._gbuf.is_cuda:      # CPU-side test backend: staged through the host
                host = self._gbuf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self._gbuf.copy_(host)
            else:
                dist.all_reduce(self._gbuf, op=dist.ReduceOp.SUM, group=group)

    def zero_grad(self):
        if self.lazy_zero:
            self._stale = True          # resolved by _resolve_stale() before the gradients are consumed
            if self._autograd_grads:
                torch._foreach_zero_(self._autograd_grads)
        else:
            self._gbuf.zero_()
        self._clear_written()

    def _resolve_stale(self):
        """lazy_zero: gradients that no kernel has overwritten since zero_grad() still hold the previous step's values --
        zero exactly those (none in a regular step)."""
        if not self._stale:
            return
        self._stale = False
        key = frozenset(id(m) for m in self._written_log)       # the same few modules every step: one dictionary lookup
        todo = self._uncovered.get(key)
        if todo is None:
            todo = self._uncovered[key] = [p for p, chain in self._covered
                                           if not any(getattr(m, '_spg_grad_written', False) for m in chain)]
        for p in todo:
            p.grad.zero_()

    def _clear_written(self):
        for m in self._written_log:
            m._spg_grad_written = False
        del self._written_log[:]

    def clamp_grad_(self, clip: float):
        self._resolve_stale()
        if clip > 0:
            self.flat.grad.clamp_(-clip, clip)

    def allreduce(self, local_weight: float = 1.0, group=None, prescaled: bool = False):
        """Weighted data-parallel mean of the gradients (see superpoint_graph_amd/dist.py), in place on the arena.
        prescaled: the loss was already multiplied by local_weight (synchronised-BatchNorm mode)."""
        from . import _lib
        self._resolve_stale()
        native = _lib.lib().spg_rccl_world_size()           # the library's own communicator (dist.init_native_rccl)
        if native <= 1 and not (dist.is_initialized() and dist.get_world_size(group) > 1):
            if prescaled:
                self.flat.grad.div_(float(local_weight))
            return
        if not prescaled:
            self.flat.grad.mul_(float(local_weight))
        self._gbuf[self.numel] = float(local_weight)
        if native > 1:            # ONE RCCL all-reduce of the arena, enqueued by the C library on the current stream
            _lib.check(_lib.lib().spg_rccl_allreduce_sum_f32(self._gbuf.data_ptr(), self.numel + 1,
                                                             torch.cuda.current_stream().cuda_stream), 'spg_rccl_allreduce_sum_f32')
        else:
            dist.all_reduce(self._gbuf, op=dist.ReduceOp.SUM, group=group)
        self.flat.grad.div_(self._gbuf[self.numel])


def _direct_writer(m):
    """Module classes whose HIP backward WRITES the gradient views of all parameters below them (PointNet incl. its STN,
    a stand-alone STNkD, the RNN-ECC module incl. filter network and cell, HipLinear on the kernel path)."""
    from .learning.modules import HipLinear, RNNGraphConvModule
    from .learning.pointnet import PointNet, STNkD
    if isinstance(m, HipLinear):
        return m._kernel_shape_ok()
    return isinstance(m, (PointNet, STNkD, RNNGraphConvModule))


def prepare_autograd_fallback(module):
    """A kernel-covered module that takes torch's path for ONE call (HipLinear on an input the kernels do not serve):
    autograd will ACCUMULATE into the arena views, so under lazy zeroing they must be cleared before that backward -- now,
    at forward time, once per zero_grad() interval -- and the module counts as written so _resolve_stale() leaves them."""
    if not getattr(module, '_spg_lazy_zero', False) or getattr(module, '_spg_grad_written', False):
        return
    for p in module.parameters(recurse=False):
        if p.grad is not None:
            p.grad.zero_()
    module._spg_grad_written = True
    log = getattr(module, '_spg_written_log', None)
    if log is not None:
        log.append(module)


def mark_direct_write(module):
    """The HIP backward kernels WRITE the gradient views of a FlatParameters model (they do not accumulate, unlike
    autograd): a second backward through the same module before `FlatParameters.zero_grad()` / the arena's optimizer step
    (gradient accumulation over several batches, one module used twice in a graph) would silently drop the first
    gradients -- raise instead."""
    if getattr(module, '_spg_grad_written', False):
        raise RuntimeError(f'{type(module).__name__}: second backward into the flat gradient arena without '
                           'FlatParameters.zero_grad() in between; the HIP kernels overwrite (do not accumulate) the '
                           'gradient views -- gradient accumulation is not supported in FlatParameters mode')
    module._spg_grad_written = True
    log = getattr(module, '_spg_written_log', None)
    if log is not None:
        log.append(module)
