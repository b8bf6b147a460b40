"""CPU, world_size 2 over gloo: the data-parallel gradient bucket (superpoint_graph_amd/dist.py) reproduces the
single-process gradient of the whole batch (weighted by the per-rank loss normalisers)."""
import os
import socket

import torch
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _make(seed=0):
    torch.manual_seed(seed)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 5))
    x = torch.randn(23, 6)
    y = torch.randint(0, 5, (23,))
    y[::7] = -100
    cw = torch.linspace(0.5, 1.5, 5)
    return model, x, y, cw


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from superpoint_graph_amd import dist as spd
    r, l, w = spd.init_from_env('gloo')
    assert (r, w) == (rank, world)
    model, x, y, cw = _make()
    cut = 9                                   # unequal shards: 9 vs 14 "superpoints"
    xs, ys = (x[:cut], y[:cut]) if rank == 0 else (x[cut:], y[cut:])
    loss = F.cross_entropy(model(xs), ys, weight=cw)
    loss.backward()
    bucket = spd.GradBucket(model.parameters())
    wtot = bucket.allreduce(spd.loss_weight(ys, cw))
    ret[rank] = ([p.grad.clone() for p in model.parameters()], float(wtot))
    torch.distributed.destroy_process_group()


def test_weighted_gradient_allreduce_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    model, x, y, cw = _make()
    F.cross_entropy(model(x), y, weight=cw).backward()
    ref = [p.grad for p in model.parameters()]
    from superpoint_graph_amd import dist as spd
    assert abs(ret[0][1] - spd.loss_weight(y, cw)) < 1e-5
    for r in (0, 1):
        for a, b in zip(ret[r][0], ref):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_bucket_world1_is_identity_and_sharding():
    from superpoint_graph_amd import dist as spd
    model, x, y, cw = _make(1)
    F.cross_entropy(model(x), y, weight=cw).backward()
    before = [p.grad.clone() for p in model.parameters()]
    spd.GradBucket(model.parameters()).allreduce(spd.loss_weight(y, cw))
    for a, b in zip(before, [p.grad for p in model.parameters()]):
        assert torch.allclose(a, b, atol=1e-7, rtol=1e-6)
    assert spd.shard_scenes(8, 3, 8) == [3] and spd.shard_scenes(10, 0, 4) == [0, 1, 2] and spd.shard_scenes(10, 3, 4) == [8, 9]
    assert sorted(sum((spd.shard_scenes(13, r, 4) for r in range(4)), [])) == list(range(13))


def _flat_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from superpoint_graph_amd import dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    spd.init_from_env('gloo')
    model, x, y, cw = _make()
    arena = FlatParameters(model)
    cut = 9
    xs, ys = (x[:cut], y[:cut]) if rank == 0 else (x[cut:], y[cut:])
    arena.zero_grad()
    F.cross_entropy(model(xs), ys, weight=cw).backward()
    arena.allreduce(spd.loss_weight(ys, cw))
    arena.clamp_grad_(0.05)
    opt = torch.optim.Adam([arena.flat], lr=1e-2)
    opt.step()
    ret[rank] = ([p.grad.clone() for p in model.parameters()], [p.detach().clone() for p in model.parameters()])
    torch.distributed.destroy_process_group()


def test_flat_parameters_world2_matches_single_process():
    """arena all-reduce + clamp + Adam over the flat buffer == reference-style per-parameter clamp + Adam on the whole batch"""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_flat_worker, args=(2, port, ret), nprocs=2, join=True)
    model, x, y, cw = _make()
    F.cross_entropy(model(x), y, weight=cw).backward()
    for p in model.parameters():
        p.grad.data.clamp_(-0.05, 0.05)                      # learning/main.py:210-212
    ref_g = [p.grad.clone() for p in model.parameters()]
    torch.optim.Adam(model.parameters(), lr=1e-2).step()
    for r in (0, 1):
        for a, b in zip(ret[r][0], ref_g):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)
        for a, b in zip(ret[r][1], model.parameters()):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def _prescaled_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from superpoint_graph_amd import dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    spd.init_from_env('gloo')
    model, x, y, cw = _make()
    arena = FlatParameters(model)
    cut = 9
    xs, ys = (x[:cut], y[:cut]) if rank == 0 else (x[cut:], y[cut:])
    arena.zero_grad()
    F.cross_entropy(model(xs), ys, weight=cw, reduction='sum').backward()       # = w_r * mean loss of the rank
    arena.allreduce(spd.loss_weight(ys, cw), prescaled=True)
    g_arena = [p.grad.clone() for p in model.parameters()]
    bucket = spd.GradBucket(model.parameters())
    model.zero_grad()
    arena.zero_grad()
    F.cross_entropy(model(xs), ys, weight=cw, reduction='sum').backward()
    bucket.allreduce(spd.loss_weight(ys, cw), prescaled=True)
    ret[rank] = (g_arena, [p.grad.clone() for p in model.parameters()])
    torch.distributed.destroy_process_group()


def test_prescaled_loss_allreduce_world2():
    """synchronised-BatchNorm mode: loss scaled by w_r before the backward, all-reduce = plain sum / w_tot"""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_prescaled_worker, args=(2, port, ret), nprocs=2, join=True)
    model, x, y, cw = _make()
    F.cross_entropy(model(x), y, weight=cw).backward()
    ref = [p.grad for p in model.parameters()]
    for r in (0, 1):
        for k in (0, 1):
            for a, b in zip(ret[r][k], ref):
                assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_flat_parameters_keeps_state_dict():
    from superpoint_graph_amd.flat import FlatParameters
    model, x, y, cw = _make(3)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    arena = FlatParameters(model)
    assert arena.numel >= sum(v.numel() for v in sd0.values()) and arena.numel % 64 == 0
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd0[k]) and v.shape == sd0[k].shape
    with torch.no_grad():
        arena.flat.add_(1.0)                                 # the module parameters are views of the arena
    for k, v in model.state_dict().items():
        assert torch.allclose(v, sd0[k] + 1.0)


def test_balanced_shards_deal_scenes_by_size():
    """dist.balanced_shards: same scene counts per rank as the contiguous blocks, every scene exactly once,
    identical on every rank (deterministic), and the slowest rank's load never above the contiguous assignment's on skewed
    sizes; equal sizes reproduce a valid partition."""
    import numpy as np
    from superpoint_graph_amd.dist import balanced_shards, shard_scenes
    rng = np.random.default_rng(0)
    for n, world in ((8, 8), (16, 8), (10, 4), (7, 2), (3, 4)):
        costs = np.clip(rng.lognormal(np.log(900.0), 0.8, n), 50, 6000)
        shards = balanced_shards(list(costs), world)
        assert shards == balanced_shards(list(costs), world)
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert [len(s) for s in shards] == [len(shard_scenes(n, r, world)) for r in range(world)]
        worst = max(sum(costs[i] for i in s) for s in shards)
        contiguous = max(sum(costs[i] for i in shard_scenes(n, r, world)) for r in range(world))
        assert worst <= contiguous + 1e-9
    assert sorted(i for s in balanced_shards([5.0] * 6, 3) for i in s) == list(range(6))
    # two scenes per rank, sizes 1..8: LPT pairs large with small
    sh = balanced_shards([1, 2, 3, 4, 5, 6, 7, 8], 4)
    assert max(sum([1, 2, 3, 4, 5, 6, 7, 8][i] for i in s) for s in sh) == 9
