"""Data-parallel layer: one process per GPU, scenes sharded one (or k) per rank, ONE flat fp32 gradient bucket
all-reduced per step with RCCL over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

The reference is single-process; this is the new multi-GPU mode BASELINE.json asks for.
Semantics: the reference's loss is a (class-weighted) mean over the labelled superpoints of the WHOLE batch
(learning/main.py:205).  With unequal per-rank counts the exact data-parallel gradient is
    g = sum_r (w_r / w_tot) * g_r,   w_r = sum of class weights of rank r's labelled superpoints,
so every rank scales its local gradients by w_r, the bucket (gradients + w_r in one extra slot) is summed by
ONE all-reduce, and the result is divided by w_tot.  The bucket is ~1.1 MB (279 409 floats for S3DIS): the
collective is latency-bound on xGMI, so a single bucket and no overlap machinery is the right shape.
BatchNorm statistics stay per-rank (local BN): each rank normalises over its own scenes.
The all-reduce happens after CloudEmbedder.bw_hook() and before the element-wise gradient clamp
(learning/main.py:208-212 order)."""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
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


def shard_scenes(n_scenes: int, rank: int, world: int):
    """Scene indices of this rank: contiguous blocks, as even as possible (scenes are independent units)."""
    base, rem = divmod(n_scenes, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


class GradBucket:
    """One flat fp32 buffer for all gradients (+1 slot for the loss weight), reused every step."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=dev)

    def allreduce(self, local_weight: float = 1.0, group=None):
        """Weighted data-parallel mean of the gradients, in place.  No-op maths at world_size 1 but the same code
        path (flatten -> [all-reduce] -> unflatten)."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
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


def loss_weight(label_mode: torch.Tensor, class_weights: Optional[torch.Tensor] = None) -> float:
    """w_r: the normaliser of this rank's cross entropy (ignore_index = -100, learning/main.py:205)."""
    valid = label_mode >= 0
    if class_weights is None:
        return float(valid.sum())
    return float(class_weights.to(label_mode.device)[label_mode[valid]].sum())
