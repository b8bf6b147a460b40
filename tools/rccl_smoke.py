#!/usr/bin/env python3
"""Single-rank smoke test of every collective the data-parallel mode issues, on the RCCL (`nccl`) backend: the calls
and tensor kinds are exactly those of bench.py / flat.py / dist.py at world_size > 1 (the build box has one GPU; real
multi-GPU runs are the driver's)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    from superpoint_graph_amd import dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    rank, local, world = spd.init_from_env(force=True)
    assert dist.get_backend() == 'nccl'
    dev = torch.device('cuda', 0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)).to(dev)
    arena = FlatParameters(model)
    model(torch.randn(5, 8, device=dev)).sum().backward()
    g0 = arena.flat.grad.clone()
    # what FlatParameters.allreduce does at world_size > 1
    arena._gbuf[arena.numel] = 3.0
    arena.flat.grad.mul_(3.0)
    dist.all_reduce(arena._gbuf, op=dist.ReduceOp.SUM)
    arena.flat.grad.div_(arena._gbuf[arena.numel])
    assert torch.allclose(arena.flat.grad, g0, rtol=1e-6, atol=1e-7)
    # bench.py: barrier + MAX over the ranks' wall times
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 1.25
    # synchronised BatchNorm: fp64 all-reduce of a prefix view of the registered buffer, from inside the C callback
    st = spd.enable_sync_bn(dev, mode='finalize')
    buf = spd._SYNC_BN['buf']
    buf[:7] = torch.arange(7, dtype=torch.float64, device=dev)
    dist.all_reduce(buf[:7], op=dist.ReduceOp.SUM)
    assert buf[:7].tolist() == list(range(7))
    spd.disable_sync_bn()
    # slot-synchronised BatchNorm: an int64 all-reduce of a zero-copy view of raw device memory, as the C callback does
    st = spd.enable_sync_bn(dev, mode='slots')
    words = torch.arange(11, dtype=torch.int64, device=dev)
    view = torch.as_tensor(spd._DevicePtr(words.data_ptr(), words.numel()), device=dev)
    dist.all_reduce(view, op=dist.ReduceOp.SUM)
    assert words.tolist() == list(range(11)) and view.data_ptr() == words.data_ptr()
    spd.disable_sync_bn()
    dist.barrier()
    dist.destroy_process_group()
    print('rccl smoke ok (backend nccl, world_size 1)')


if __name__ == '__main__':
    main()
