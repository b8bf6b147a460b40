"""Data-parallel layer: one process per GPU, scenes sharded one (or k) per rank, ONE flat fp32 gradient bucket
all-reduced per step with RCCL over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference is single-process; this is the new multi-GPU mode BASELINE.json asks for.
Semantics: the reference's loss is a (class-weighted) mean over the labelled superpoints of the WHOLE batch
(learning/main.py:205).  With unequal per-rank counts the exact data-parallel gradient is
    g = sum_r (w_r / w_tot) * g_r,   w_r = sum of class weights of rank r's labelled superpoints,
so every rank scales its local gradients by w_r, the bucket (gradients + w_r in one extra slot) is summed by
ONE all-reduce, and the result is divided by w_tot.  The bucket is ~1.1 MB (279 409 floats for S3DIS): the
collective is latency-bound on xGMI, so a single bucket and no overlap machinery is the right shape.
BatchNorm: two modes.  Default = per-rank statistics (each rank normalises over its own scenes; what plain data
parallelism gives).  `enable_sync_bn()` = statistics over the union of all ranks' scenes, which reproduces the
reference's single-process batch exactly: the 13 train-mode BatchNorm layers all-reduce their
per-channel fp64 sums (forward: 3C+1 doubles, backward: 2C+1) through the callback the C library exposes
(include/spg_hip.h: spg_set_bn_allreduce).  In that mode the backward couples the ranks, so the loss has to be
scaled BEFORE the backward: loss_r = w_r * CE_r, then `allreduce(w_r, prescaled=True)` sums and divides by w_tot.
The gradient all-reduce happens after CloudEmbedder.bw_hook() and before the element-wise gradient clamp
(learning/main.py:208-212 order)."""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world_size).
    force: create the process group even for a single rank (smoke test of the RCCL path on a 1-GPU box)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def init_native_rccl(group=None):
    """Gives libspg_hip.so a RCCL communicator of its own over the ranks of `group` (include/spg_hip.h: spg_rccl_*): rank 0
    draws the unique id, the 128 bytes are broadcast through torch.distributed (any backend), every rank initialises
    with its CURRENT device.  Afterwards the flat gradient all-reduce (`FlatParameters.allreduce`) and the synchronised
    BatchNorm all-reduces are enqueued by the C library itself on torch's current stream -- no Python callback, no
    torch collective on the data path.  Works at world size 1 (smoke test of the native path on a 1-GPU box)."""
    import ctypes
    from ._lib import check, lib
    L = lib()
    if L.spg_rccl_world_size() > 0:
        return L.spg_rccl_world_size()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    raw = (ctypes.c_ubyte * 128)()
    if rank == 0:
        check(L.spg_rccl_unique_id(ctypes.cast(raw, ctypes.c_void_p)), 'spg_rccl_unique_id')
    ident = torch.tensor(list(raw), dtype=torch.uint8)
    if world > 1:
        on_gpu = dist.get_backend(group) == 'nccl'
        ident = ident.cuda() if on_gpu else ident
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ident.cpu()
    raw = (ctypes.c_ubyte * 128)(*ident.tolist())
    check(L.spg_rccl_init(ctypes.cast(raw, ctypes.c_void_p), world, rank), 'spg_rccl_init')
    return world


def native_rccl_world_size() -> int:
    from ._lib import lib
    return int(lib().spg_rccl_world_size())


def shard_scenes(n_scenes: int, rank: int, world: int):
    """Scene indices of this rank: contiguous blocks, as even as possible (scenes are independent units)."""
    base, rem = divmod(n_scenes, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def balanced_shards(costs, world: int):
    """Scene positions of EVERY rank for one global batch whose scenes cost `costs` (superpoints per scene): the same scene
    COUNTS as shard_scenes (ceil / floor of n / world -- memory and the per-rank BatchNorm batch stay comparable), but which
    scene goes where is chosen greedily, largest first, to the least-loaded rank that still has a free slot (LPT).  A step takes
    as long as its slowest rank (bench.py: max over ranks), real scenes differ several-fold in size, and contiguous blocks
    leave that to chance.  Deterministic (ties by position), so every rank computes the same assignment without communication.
    -> list of `world` lists of positions into `costs`."""
    n = len(costs)
    base, rem = divmod(n, world)
    slots = [base + (1 if r < rem else 0) for r in range(world)]
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for pos in sorted(range(n), key=lambda i: (-float(costs[i]), i)):
        r = min((r for r in range(world) if len(out[r]) < slots[r]), key=lambda r: (load[r], r))
        out[r].append(pos)
        load[r] += float(costs[pos])
    return [sorted(o) for o in out]


class GradBucket:
    """One flat fp32 buffer for all gradients (+1 slot for the loss weight), reused every step."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=dev)

    def allreduce(self, local_weight: float = 1.0, group=None, prescaled: bool = False):
        """Weighted data-parallel mean of the gradients, in place.  No-op maths at world_size 1 but the same code
        path (flatten -> [all-reduce] -> unflatten).  prescaled: the loss was already multiplied by local_weight
        (synchronised-BatchNorm mode)."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if not prescaled:
            self.flat[:self.numel].mul_(float(local_weight))
        self.flat[self.numel] = float(local_weight)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat[:self.numel].div_(self.flat[self.numel])
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
        return self.flat[self.numel]


_SYNC_BN = {}


class _DevicePtr:
    """zero-copy torch view of `n` int64 words of device memory (the statistics slots inside a caller's workspace tensor)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<i8', 'data': (int(ptr), False), 'version': 3, 'strides': None}


def sync_bn_mode():
    """None (per-rank statistics), 'slots' or 'finalize'"""
    return _SYNC_BN.get('mode')


def _enable_slot_sync(device, group=None):
    """Slot-synchronised BatchNorm (include/spg_hip.h: spg_set_slot_allreduce): the ranks all-reduce the exact fixed-point
    statistics slots themselves (int64 sums) between producer and consumer launch -- synchronised statistics at the speed of the
    per-rank mode (statistics folds, fused convolution backward, one-pass first layers, spg_train_step all keep running), and
    bit-identical on every rank.  The consumers' row counts travel inside the slots (a spare word counts the producers' rows), so
    there is nothing else to exchange."""
    import ctypes
    from ._lib import check, lib
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if lib().spg_rccl_world_size() > 0:
        check(lib().spg_rccl_sync_slots(1), 'spg_rccl_sync_slots')
        state = {'calls': None, 'error': None, 'native': True, 'mode': 'slots'}
        _SYNC_BN.update(cb=None, state=state, mode='slots', group=group, native=True)
        return state
    staged = dist.is_initialized() and dist.get_backend(group) == 'gloo'
    state = {'calls': 0, 'error': None, 'mode': 'slots'}

    def _allreduce(ctx, ptr, n, stream):
        try:
            state['calls'] += 1
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                view = torch.as_tensor(_DevicePtr(ptr, n), device=device)
                if staged:                         # CPU-side test backend: through the host (synchronises)
                    host = view.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                    view.copy_(host)
                else:                              # RCCL: enqueued in stream order w.r.t. torch's current stream
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:                     # never let an exception cross the C boundary
            state['error'] = e
            return 1

    cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p)(_allreduce)
    check(lib().spg_set_slot_allreduce(ctypes.cast(cb, ctypes.c_void_p), None, world), 'spg_set_slot_allreduce')
    _SYNC_BN.update(cb=cb, state=state, mode='slots', group=group, native=False)
    return state


def enable_sync_bn(device, group=None, max_channels: int = 1024, mode=None):
    """Synchronise the BatchNorm statistics of the HIP path over `group` (all ranks must run the same layers).
    mode 'slots' (default, round 5): the ranks all-reduce the exact fixed-point statistics slots -- see _enable_slot_sync;
    mode 'finalize' (rounds 1-4): every BatchNorm layer reduces into a finalize launch whose fp64 sums are all-reduced (no
    statistics folds, no fused convolution backward, no spg_train_step).  SPG_SYNC_BN_MODE overrides the default.
    With the library's own communicator (`init_native_rccl`) the all-reduces are issued by the C library directly; otherwise
    through a callback into torch.distributed (gloo staging for the CPU-side tests)."""
    import ctypes
    from ._lib import check, lib
    mode = mode or os.environ.get('SPG_SYNC_BN_MODE', 'slots')
    if mode not in ('slots', 'finalize'):
        raise ValueError("sync-BN mode must be 'slots' or 'finalize'")
    if _SYNC_BN:
        disable_sync_bn()
    if mode == 'slots':
        return _enable_slot_sync(device, group)
    buf = torch.zeros(3 * max_channels + 16, dtype=torch.float64, device=device)
    if lib().spg_rccl_world_size() > 0:
        check(lib().spg_rccl_sync_bn(buf.data_ptr(), buf.numel()), 'spg_rccl_sync_bn')
        state = {'calls': None, 'error': None, 'native': True, 'mode': 'finalize'}
        _SYNC_BN.update(cb=None, buf=buf, state=state, mode='finalize')
        return state
    staged = dist.is_initialized() and dist.get_backend(group) == 'gloo' and buf.is_cuda
    state = {'calls': 0, 'error': None, 'mode': 'finalize'}

    def _allreduce(ctx, ptr, n, stream):
        try:
            state['calls'] += 1
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                view = buf[:n]
                if staged:                         # CPU-side test backend: through the host (synchronises)
                    host = view.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                    view.copy_(host)
                else:                              # RCCL: enqueued in stream order w.r.t. torch's current stream
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:                     # never let an exception cross the C boundary
            state['error'] = e
            return 1

    cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p)(_allreduce)
    check(lib().spg_set_bn_allreduce(ctypes.cast(cb, ctypes.c_void_p), None, buf.data_ptr(), buf.numel()),
          'spg_set_bn_allreduce')
    _SYNC_BN.update(cb=cb, buf=buf, state=state, mode='finalize')
    return state


def disable_sync_bn():
    from ._lib import check, lib
    check(lib().spg_set_bn_allreduce(None, None, None, 0), 'spg_set_bn_allreduce')
    check(lib().spg_set_slot_allreduce(None, None, 1), 'spg_set_slot_allreduce')
    _SYNC_BN.clear()


def loss_weight(label_mode: torch.Tensor, class_weights: Optional[torch.Tensor] = None) -> float:
    """w_r: the normaliser of this rank's cross entropy (ignore_index = -100, learning/main.py:205)."""
    valid = label_mode >= 0
    if class_weights is None:
        return float(valid.sum())
    return float(class_weights.to(label_mode.device)[label_mode[valid]].sum())
