"""The persistent, dataflow-synchronised form of the GRU recurrence (spg_ecc_persist_{fwd,bwd}_kernel: all iterations in one
launch, neighbour states exchanged as tagged granules) against the per-iteration launches it replaces (spg_tune key 8): the two
forms execute the same arithmetic in the same order, so outputs and every gradient must be BIT-IDENTICAL -- a stale or torn
hand-off shows up as a difference.  Graph shapes: the BASELINE scene size, hubs with more in-/out-edges than the
register-resident filters (8) and than one gather pass (32), isolated nodes, a partial last workgroup; repeated launches with the
GPU busy in between (uneven load), and the time-out counter must stay 0.  The oracle comparison of the same module is
tests/test_gpu_model.py::test_rnn_ecc_module_large_graph_vs_oracle (7000 nodes: per-iteration path) and the golden tests
(<= 49 nodes: persistent path)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _graph(n, e, seed, hubs=True):
    rng = np.random.default_rng(seed)
    tgt = rng.integers(0, n, size=e)
    src = rng.integers(0, n, size=e)
    if hubs and n > 100:
        tgt[:70] = 3                      # in-degree >= 70: more than two gather passes
        src[70:150] = 5                   # out-degree >= 80
        tgt[150:162] = 9                  # between the resident filters (8) and one pass (32)
        tgt[tgt == 11] = 12               # node 11: no in-edges
        src[src == 13] = 14               # node 13: no out-edges
    order = np.argsort(tgt, kind='stable')
    tgt, src = tgt[order], src[order]
    idxn = torch.from_numpy(src.astype(np.int64))
    degs = torch.from_numpy(np.bincount(tgt, minlength=n).astype(np.int64))
    return idxn, degs


def _run(net, gi, x, go, legacy):
    from superpoint_graph_amd import _lib
    L = _lib.lib()
    old = L.spg_tune(8, 1 if legacy else 0)
    try:
        net.zero_grad()
        net.set_info([gi], 1)
        xg = x.clone().requires_grad_(True)
        out = net(xg)
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
    finally:
        L.spg_tune(8, old)


@pytest.mark.parametrize('config,n,e', [('gru_10_0,f_13', 1000, 5000), ('gru_10_0,f_13', 1003, 6000), ('gru_3_0_1_1_0,f_5', 37, 150),
                                        ('gru_4_1,f_8', 1000, 5000), ('gru_2_1_0_0,f_8', 130, 700),
                                        # one component above 2048 nodes (round 5): several nodes per wavefront, iteration-major
                                        ('gru_10_0,f_13', 2100, 9000), ('gru_10_0,f_13', 5000, 25000), ('gru_4_1,f_8', 10000, 50000),
                                        ('gru_3_0_1_1_0,f_5', 10000, 50000)])
def test_persistent_recurrence_is_bit_identical_to_per_iteration_launches(hip, config, n, e):
    from superpoint_graph_amd.learning import ecc, graphnet
    idxn, degs = _graph(n, e, seed=n + e)
    edgefeats = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(2)).to(DEV)
    torch.manual_seed(7)
    net = graphnet.GraphNetwork(config, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).to(DEV).train()
    gi = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None)
    with torch.no_grad():
        net.set_info([gi], 1)
        width = net(x).shape[1]
    go = torch.randn(n, width, generator=torch.Generator().manual_seed(3)).to(DEV)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    ref = _run(net, gi, x, go, legacy=True)
    busy = torch.randn(4096, 4096, device=DEV)
    for rep in range(6):
        net.load_state_dict(state0)
        if rep % 2:                       # uneven load: a large GEMM is in flight when the persistent launches start
            busy = busy @ busy * 1e-3
        got = _run(net, gi, x, go, legacy=False)
        assert torch.equal(got[0], ref[0]), f'forward differs (repetition {rep})'
        assert torch.equal(got[1], ref[1]), f'input gradient differs (repetition {rep})'
        for k in ref[2]:
            assert torch.equal(got[2][k], ref[2][k]), f'{k} differs (repetition {rep})'
    assert hip.spg_ecc_persistent_errors() == 0


def test_persistent_recurrence_eval_mode_and_above_the_node_limit(hip):
    """Inference (no aggregates kept) through the persistent launch: 1000 nodes with one workgroup per CU, 1100 nodes with two
    (since round 4; up to 2048 nodes per round) -- the same results as the per-iteration launches either way."""
    from superpoint_graph_amd.learning import ecc, graphnet
    for n, e in ((1000, 5000), (1100, 5000)):
        idxn, degs = _graph(n, e, seed=n)
        edgefeats = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
        x = torch.randn(n, 32, generator=torch.Generator().manual_seed(2)).to(DEV)
        torch.manual_seed(7)
        net = graphnet.GraphNetwork('gru_10_0,f_13', 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).to(DEV).eval()
        gi = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None)
        outs = []
        for legacy in (True, False):
            old = hip.spg_tune(8, int(legacy))
            try:
                with torch.no_grad():
                    net.set_info([gi], 1)
                    outs.append(net(x).clone())
            finally:
                hip.spg_tune(8, old)
        assert torch.equal(outs[0], outs[1])
    assert hip.spg_ecc_persistent_errors() == 0


def _multi_scene(sizes, edges_per_node, seed):
    """A batch of disjoint scenes (no edge crosses a scene boundary), edges sorted by target: (idxn, degs, parts)."""
    rng = np.random.default_rng(seed)
    src, tgt, off, parts = [], [], 0, [0]
    for n in sizes:
        e = int(n * edges_per_node)
        t = rng.integers(0, n, size=e); s_ = rng.integers(0, n, size=e)
        if n > 200:
            t[:40] = 7                     # a hub above the register-resident filters and one gather pass
        src.append(s_ + off); tgt.append(t + off)
        off += n
        parts.append(off)
    src, tgt = np.concatenate(src), np.concatenate(tgt)
    order = np.argsort(tgt, kind='stable')
    return torch.from_numpy(src[order].astype(np.int64)), torch.from_numpy(np.bincount(tgt, minlength=off).astype(np.int64)), parts


@pytest.mark.parametrize('config,sizes', [('gru_10_0,f_13', [1000, 1000]),                 # 2 scenes: one round, two workgroups per CU
                                          ('gru_10_0,f_13', [1000] * 5),                   # 3 rounds of <= 2048 nodes
                                          ('gru_4_1,f_8', [700, 900, 300, 1200, 50]),      # vector filters, ragged scenes
                                          ('gru_3_0_1_1_0,f_5', [400, 300, 200, 100]),     # 1000 nodes in 4 scenes: still one round
                                          ('gru_4_1,f_8', [800] * 20)])                    # 10 rounds > 8: ONE group, several nodes per wavefront (round 5)
def test_persistent_recurrence_multi_scene_batches(hip, config, sizes):
    """Batches of several scenes: up to 2048 nodes run as ONE round with two workgroups per CU, larger batches
    in rounds of whole scenes (the scene boundaries travel in spg_eccrnn_cfg.n_parts) -- bit-identical to the per-iteration
    launches, forward, input gradient and every parameter gradient; the time-out counter stays 0."""
    from superpoint_graph_amd.learning import ecc, graphnet
    idxn, degs, parts = _multi_scene(sizes, 5, seed=sum(sizes))
    n, e = int(degs.numel()), int(idxn.numel())
    edgefeats = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(2)).to(DEV)
    torch.manual_seed(7)
    net = graphnet.GraphNetwork(config, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).to(DEV).train()
    gi = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None, parts=parts)
    with torch.no_grad():
        net.set_info([gi], 1)
        width = net(x).shape[1]
    go = torch.randn(n, width, generator=torch.Generator().manual_seed(3)).to(DEV)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    ref = _run(net, gi, x, go, legacy=True)
    for rep in range(3):
        net.load_state_dict(state0)
        got = _run(net, gi, x, go, legacy=False)
        assert torch.equal(got[0], ref[0]), f'forward differs (repetition {rep})'
        assert torch.equal(got[1], ref[1]), f'input gradient differs (repetition {rep})'
        for k in ref[2]:
            assert torch.equal(got[2][k], ref[2][k]), f'{k} differs (repetition {rep})'
    assert hip.spg_ecc_persistent_errors() == 0
    # without the scene boundaries a graph above one round takes the per-iteration launches: same results
    if n > 2048:
        gi2 = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None)
        net.load_state_dict(state0)
        got = _run(net, gi2, x, go, legacy=False)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
