"""GPU parity tests of the individual HIP operators, called through the C ABI (superpoint_graph_amd.ops /
the learning.* module API) and checked against the oracle and the golden vectors of the reference."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, maxrel
from oracle import spg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _random_graph(n, e, seed, zero_deg=True):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, e)
    dst = np.sort(rng.integers(0, n, e))
    if zero_deg:
        dst[dst == 3 % n] = (4 % n)           # node 3 gets no in-edges
        dst = np.sort(dst)
    degs = np.bincount(dst, minlength=n).astype(np.int64)
    return torch.from_numpy(src.astype(np.int64)), torch.from_numpy(degs)


def test_graph_build_bit_exact(hip):
    from superpoint_graph_amd import ops
    for n, e, seed in ((5, 50, 0), (1000, 5000, 1), (10000, 50000, 2), (7, 0, 3)):
        idxn, degs = _random_graph(n, e, seed) if e else (torch.zeros(0, dtype=torch.long), torch.zeros(n, dtype=torch.long))
        g = ops.DeviceGraph(idxn.to(DEV), degs.to(DEV))
        rowptr, src, dst, rrp, rev = [t.cpu().numpy() for t in g.export()]
        assert g.hdr.cpu().tolist() == [n, n, e, 0]
        assert np.array_equal(rowptr, O.csr_by_target(degs.numpy()))
        assert np.array_equal(src, idxn.numpy()) and np.array_equal(dst, O.edge_targets(degs.numpy()))
        rp, order = O.csr_by_source(idxn.numpy(), n)
        assert np.array_equal(rrp, rp) and np.array_equal(rev, order)
    # more input rows than output nodes (the reference's test fixture) and a malformed index buffer
    idxn, degs = torch.tensor([7, 0, 19, 3, 3]), torch.tensor([2, 0, 3])
    g = ops.DeviceGraph(idxn.to(DEV), degs.to(DEV), n_src=20)
    rowptr, src, dst, rrp, rev = [t.cpu().numpy() for t in g.export()]
    rp, order = O.csr_by_source(idxn.numpy(), 20)
    assert np.array_equal(rrp, rp) and np.array_equal(rev, order) and g.hdr.cpu().tolist() == [3, 20, 5, 0]
    g = ops.DeviceGraph(idxn.to(DEV), degs.to(DEV))            # n_src defaults to N=3: indices 7, 19 are out of range
    g.export()
    assert g.hdr.cpu().tolist()[3] == 1


def test_generic_ecc_fp64_golden_and_gradcheck(hip):
    from superpoint_graph_amd.learning import ecc
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    x, w = torch.from_numpy(g['ecc_x']).to(DEV), torch.from_numpy(g['ecc_w']).to(DEV)
    idxn, degs = torch.from_numpy(g['ecc_idxn']).to(DEV), torch.from_numpy(g['ecc_degs'])
    out = ecc.GraphConvFunction.apply(x, w, 10, 15, idxn, None, degs, degs.to(DEV), 30)
    assert maxrel(out, torch.from_numpy(g['ecc_out'])) < 1e-13
    assert float(out[1].abs().max()) == 0.0
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    o = ecc.GraphConvFunction.apply(xg, wg, 10, 15, idxn, None, degs, degs.to(DEV), 1)
    o.backward(torch.ones_like(o))
    assert maxrel(xg.grad, torch.from_numpy(g['ecc_gx_ones'])) < 1e-13 and maxrel(wg.grad, torch.from_numpy(g['ecc_gw_ones'])) < 1e-13
    # the reference's own test: fp64 gradcheck, with and without filter sharing (test_GraphConvModule.py:23-57)
    dg = degs.to(DEV)
    assert torch.autograd.gradcheck(lambda a, b: ecc.GraphConvFunction.apply(a, b, 10, 15, idxn, None, degs, dg, 30), (xg, wg))
    gen = torch.Generator().manual_seed(1)
    idxe = torch.randint(0, 30, (50,), generator=gen).to(DEV)
    w30 = torch.randn(30, 10, 15, generator=gen, dtype=torch.float64).to(DEV).requires_grad_(True)
    # filter sharing accumulates grad_weights with fp64 atomics (order-dependent in the last bits): nondet_tol
    assert torch.autograd.gradcheck(lambda a, b: ecc.GraphConvFunction.apply(a, b, 10, 15, idxn, idxe, degs, dg, 30), (xg, w30),
                                    nondet_tol=1e-10)
    # vector filters
    xv, wv = torch.from_numpy(g['eccv_x']).to(DEV).requires_grad_(True), torch.from_numpy(g['eccv_w']).to(DEV).requires_grad_(True)
    ov = ecc.GraphConvFunction.apply(xv, wv, 12, 12, idxn, None, degs, dg, 1e10)
    assert maxrel(ov, torch.from_numpy(g['eccv_out'])) < 1e-13
    ov.backward(torch.from_numpy(g['eccv_go']).to(DEV))
    assert maxrel(xv.grad, torch.from_numpy(g['eccv_gx'])) < 1e-13 and maxrel(wv.grad, torch.from_numpy(g['eccv_gw'])) < 1e-13


@pytest.mark.parametrize('matrix', [True, False])
@pytest.mark.parametrize('n,e', [(40, 150), (1000, 5000), (7000, 30000)])
def test_fused_ecc_32_channels(hip, matrix, n, e):
    """hot-path shape through the fused kernels of spg_ecc_aggregate_fwd (wave per node) and spg_ecc_aggregate_bwd (wave per
    source node over the reverse CSR for grad_x, wave per edge for grad_w) against the fp64 oracle."""
    from superpoint_graph_amd.learning import ecc
    idxn, degs = _random_graph(n, e, 7)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(n, 32, generator=gen)
    w = torch.randn(e, 32, 32, generator=gen) if matrix else torch.randn(e, 32, generator=gen)
    go = torch.randn(n, 32, generator=gen)
    ref = O.ecc_forward(x.double(), w.double(), idxn, degs)
    gx_ref, gw_ref = O.ecc_backward(x.double(), w.double(), go.double(), idxn, degs)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    out = ecc.GraphConvFunction.apply(xg, wg, 32, 32, idxn.to(DEV), None, degs, degs.to(DEV), 30000)
    assert maxrel(out, ref) < 2e-6
    assert float(out[3].abs().max()) == 0.0
    out.backward(go.to(DEV))
    assert maxrel(xg.grad, gx_ref) < 2e-6 and maxrel(wg.grad, gw_ref) < 2e-6


@pytest.mark.parametrize('layernorm,ingate', [(True, True), (False, False), (True, False)])
def test_gru_cell(hip, layernorm, ingate):
    from superpoint_graph_amd.learning import modules
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    torch.manual_seed(5)
    cell = modules.GRUCellEx(32, 32, bias=True, layernorm=layernorm, ingate=ingate)
    if layernorm and ingate:
        cell.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('gru_p/')})
    cell = cell.to(DEV)
    gen = torch.Generator().manual_seed(3)
    n = 9 if (layernorm and ingate) else (3333 if ingate or layernorm else 7001)
    inp = torch.from_numpy(g['gru_in']) if n == 9 else torch.randn(n, 32, generator=gen)
    hid = torch.from_numpy(g['gru_h']) if n == 9 else torch.randn(n, 32, generator=gen)
    xi, xh = inp.to(DEV).requires_grad_(True), hid.to(DEV).requires_grad_(True)
    out = cell(xi, xh)
    if n == 9:
        assert maxrel(out, torch.from_numpy(g['gru_out'])) < 2e-6        # the reference's own output
    # fp64 oracle forward + autograd backward
    P = {'c.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in cell.state_dict().items()}
    oi, oh = inp.double().requires_grad_(True), hid.double().requires_grad_(True)
    ref = O.gru_cell_ex(oi, oh, P, 'c', layernorm, ingate)
    assert maxrel(out, ref) < 2e-6
    go = torch.randn(n, 32, generator=gen)
    out.backward(go.to(DEV))
    ref.backward(go.double())
    assert maxrel(xi.grad, oi.grad) < 5e-6 and maxrel(xh.grad, oh.grad) < 5e-6
    for k, p in cell.named_parameters():
        assert maxrel(p.grad, P['c.' + k].grad) < 5e-6, k


@pytest.mark.parametrize('layernorm,ingate', [(True, True), (False, False), (True, False), (False, True)])
def test_lstm_cell(hip, layernorm, ingate):
    """LSTMCellEx.forward (reference learning/modules.py:280-309): the reference's own outputs on the golden inputs, then
    the fp64 oracle + autograd for every gradient (incl. the one entering through cy)."""
    from superpoint_graph_amd.learning import modules
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    torch.manual_seed(6)
    cell = modules.LSTMCellEx(32, 32, bias=True, layernorm=layernorm, ingate=ingate)
    if layernorm and ingate:
        cell.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('lstm_p/')})
    cell = cell.to(DEV)
    gen = torch.Generator().manual_seed(4)
    n = 9 if (layernorm and ingate) else (257 if layernorm else 6500)
    inp = torch.from_numpy(g['gru_in']) if n == 9 else torch.randn(n, 32, generator=gen)
    hid = torch.from_numpy(g['gru_h']) if n == 9 else torch.randn(n, 32, generator=gen)
    cx = torch.from_numpy(g['lstm_c']) if n == 9 else torch.randn(n, 32, generator=gen)
    xi, xh, xc = [t.to(DEV).requires_grad_(True) for t in (inp, hid, cx)]
    hy, cy = cell(xi, (xh, xc))
    if n == 9:
        assert maxrel(hy, torch.from_numpy(g['lstm_hy'])) < 2e-6        # the reference's own outputs
        assert maxrel(cy, torch.from_numpy(g['lstm_cy'])) < 2e-6
    P = {'c.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in cell.state_dict().items()}
    oi, oh, oc = [t.double().requires_grad_(True) for t in (inp, hid, cx)]
    rhy, rcy = O.lstm_cell_ex(oi, (oh, oc), P, 'c', layernorm, ingate)
    assert maxrel(hy, rhy) < 2e-6 and maxrel(cy, rcy) < 2e-6
    gh, gc = torch.randn(n, 32, generator=gen), torch.randn(n, 32, generator=gen)
    torch.autograd.backward([hy, cy], [gh.to(DEV), gc.to(DEV)])
    torch.autograd.backward([rhy, rcy], [gh.double(), gc.double()])
    for a, b in ((xi, oi), (xh, oh), (xc, oc)):
        assert maxrel(a.grad, b.grad) < 5e-6
    for k, p in cell.named_parameters():
        assert maxrel(p.grad, P['c.' + k].grad) < 5e-6, k
    # only hy used downstream: the cy gradient is absent
    cell.zero_grad()
    hy2, _ = cell(xi.detach().requires_grad_(True), (xh.detach(), xc.detach()))
    hy2.sum().backward()
    P2 = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    r2, _ = O.lstm_cell_ex(inp.double(), (hid.double(), cx.double()), P2, 'c', layernorm, ingate)
    r2.sum().backward()
    assert maxrel(cell.weight_hh.grad, P2['c.weight_hh'].grad) < 5e-6


@pytest.mark.parametrize('M,K,N', [(300, 14, 64), (129, 13, 32), (1000, 64, 64), (517, 64, 128), (256, 128, 256),
                                   (300, 257, 256), (100, 64, 4), (640, 64, 1024), (77, 352, 13), (128, 32, 96)])
def test_linear_forward_and_wgrad(hip, M, K, N):
    from superpoint_graph_amd import ops
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    sc, sh = torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen) * 0.3
    sc[0] = -sc[0]
    dy = torch.randn(M, N, generator=gen)
    X, W, Bv, SC, SH, DY = [t.to(DEV) for t in (x, w, b, sc, sh, dy)]
    y = ops.linear_fwd(X, W, Bv)
    assert maxrel(y, x.double() @ w.double().t() + b.double()) < 2e-6
    a = torch.relu(x.double() * sc.double() + sh.double())
    y2 = ops.linear_fwd(X, W, None, SC, SH, True)
    assert maxrel(y2, a @ w.double().t()) < 2e-6
    dw = ops.linear_wgrad(DY, X)
    assert maxrel(dw, dy.double().t() @ x.double()) < 2e-6
    dw2 = ops.linear_wgrad(DY, X, SC, SH, True)
    assert maxrel(dw2, dy.double().t() @ a) < 2e-6
    # weight and bias gradient together (column sums of dY ride along with the weight-gradient launch)
    dw3, db3 = ops.linear_wgrad_bias(DY, X)
    assert torch.equal(dw3, dw)
    assert maxrel(db3, dy.double().sum(0)) < 2e-6


def test_fused_clamp_adam_matches_torch(hip):
    """spg_adam_clamp_step == per-parameter clamp_ + torch.optim.Adam (learning/main.py:210-213, :433-437)."""
    from superpoint_graph_amd.flat import FlatParameters
    torch.manual_seed(0)
    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.Tanh(), torch.nn.Linear(64, 7)).to(DEV)
    ref, mod = make(), make()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-3)
    arena = FlatParameters(mod)
    x = torch.randn(50, 20, device=DEV)
    y = torch.randint(0, 7, (50,), device=DEV)
    for step in range(4):
        opt.zero_grad()
        (torch.nn.functional.cross_entropy(ref(x), y) * 30).backward()
        for p in ref.parameters():
            p.grad.data.clamp_(-0.5, 0.5)
        opt.step()
        arena.zero_grad()
        (torch.nn.functional.cross_entropy(mod(x), y) * 30).backward()
        arena.adam_step(lr=1e-2, weight_decay=1e-3, grad_clip=0.5)
        for a, b in zip(mod.parameters(), ref.parameters()):
            assert maxrel(a.grad, b.grad) < 2e-5 and maxrel(a, b) < 2e-5, step   # fp32 trajectories drift by rounding


def test_lazy_zero_arena_with_autograd_parameters(hip):
    """FlatParameters(lazy_zero=True) -- the CLI default -- with parameters NO HIP kernel writes: the affine BatchNorm of a
    `b` token and an `f_K` layer whose input width is no multiple of 4 (torch's F.linear) get their gradients from autograd's
    AccumulateGrad, which ADDS.  They must be zeroed every step (they were accumulated onto stale values and then
    cleared right before Adam, i.e. never trained).  Trajectory against torch.optim.Adam on an un-flattened twin."""
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.learning import graphnet
    def make():
        torch.manual_seed(5)
        return graphnet.GraphNetwork('f_30,b,r,f_8', 32, [13, 32], use_pyg=0).to(DEV).train()
    ref, mod = make(), make()
    assert not dict(mod.named_children())['3']._kernel_shape_ok() and dict(mod.named_children())['0']._kernel_shape_ok()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    arena = FlatParameters(mod, lazy_zero=True, host_counters=True)
    assert len(arena._autograd_grads) == 4          # BatchNorm weight / bias, second Linear weight / bias
    x = torch.randn(64, 32, device=DEV)
    y = torch.randint(0, 8, (64,), device=DEV)
    for step in range(4):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(ref(x), y).backward()
        opt.step()
        arena.zero_grad()
        torch.nn.functional.cross_entropy(mod(x), y).backward()
        g_before = {k: p.grad.clone() for k, p in mod.named_parameters()}
        arena.adam_step(lr=1e-2)
        for (k, a), b in zip(mod.named_parameters(), ref.parameters()):
            if k == '0.bias':       # the bias in front of the train-mode BatchNorm: analytically zero gradient, round-off on both sides
                assert float(a.grad.abs().max()) < 1e-6 and float(b.grad.abs().max()) < 1e-6
                continue
            assert float(g_before[k].abs().max()) > 0, (k, step)
            assert maxrel(a.grad, b.grad) < 1e-4 and maxrel(a, b) < 1e-4, (k, step)
    # a kernel-covered HipLinear that takes torch's path for one call (3-d input): cleared at forward time, then accumulated once
    arena.zero_grad()
    fc0 = dict(mod.named_children())['0']
    xx = torch.randn(2, 5, 32, device=DEV)
    stale = fc0.weight.grad.clone()
    fc0(xx).sum().backward()
    expect = xx.reshape(-1, 32).sum(0).expand(30, 32)
    arena.adam_step(lr=0.0)
    assert maxrel(fc0.weight.grad, expect) < 1e-5 and float(stale.abs().max()) > 0


@pytest.mark.parametrize('n,c,weighted,reduction', [(1000, 13, False, 'mean'), (1000, 13, True, 'mean'), (7, 8, True, 'sum'), (4099, 13, True, 'mean')])
def test_cross_entropy_matches_torch(hip, n, c, weighted, reduction):
    """ops.cross_entropy (one launch each way) against torch.nn.functional.cross_entropy: loss, gradient, ignore_index."""
    import torch.nn.functional as F
    from superpoint_graph_amd import ops
    g = torch.Generator().manual_seed(n)
    logits = (torch.randn(n, c, generator=g) * 3).cuda().requires_grad_(True)
    target = torch.randint(0, c, (n,), generator=g)
    target[torch.rand(n, generator=g) < 0.1] = -100
    target = target.cuda()
    w = (torch.rand(c, generator=g) + 0.5).cuda() if weighted else None
    loss = ops.cross_entropy(logits, target, weight=w, reduction=reduction)
    (loss * 1.7).backward()
    g_ours = logits.grad.clone()
    logits.grad = None
    ref = F.cross_entropy(logits, target, weight=w, reduction=reduction)
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((g_ours - logits.grad).abs().max()) <= 2e-6 * float(logits.grad.abs().max())
    assert float(g_ours[target == -100].abs().max()) == 0.0


def test_backward_after_eval_forward_fails_loudly(hip):
    """The backward kernels implement batch-statistics BatchNorm: after an eval-mode forward (frozen-BN fine-tuning,
    saliency) they must refuse instead of reading the training-layout workspace out of bounds."""
    import types
    from conftest import build_model, load_golden
    from superpoint_graph_amd.learning import ecc, pointnet
    spec, batch, state0, g = load_golden('vector_gru4_small')
    model = build_model(spec, state0).cuda().eval()
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    emb = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=0)).run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
    out = model.ecc(emb)
    with pytest.raises(RuntimeError, match='eval-mode forward'):
        out.sum().backward()


def test_graph_index_validation(hip):
    """GraphConvInfo.cuda() checks the index contract on the host before the device CSR is built from it."""
    from superpoint_graph_amd.learning import ecc
    idxn = torch.tensor([0, 1, 2, 1], dtype=torch.int64)
    degs = torch.tensor([2, 1, 1], dtype=torch.int64)
    ef = torch.zeros(4, 13)
    ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), ef.clone()).cuda()                      # consistent: fine
    with pytest.raises(ValueError, match='does not match the number of edges'):
        ecc.GraphConvInfo.from_buffers(idxn.clone(), torch.tensor([2, 1, 2]), ef.clone()).cuda()
    with pytest.raises(IndexError, match='idxn entries'):
        ecc.GraphConvInfo.from_buffers(torch.tensor([0, 1, 7, 1]), degs.clone(), ef.clone()).cuda()
    with pytest.raises(ValueError, match='edge-feature rows'):
        ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), torch.zeros(3, 13)).cuda()


def test_batchnorm_option_checks(hip):
    """BatchNorm variants the kernels do not implement are refused, not silently replaced."""
    from superpoint_graph_amd.learning import pointnet
    net = pointnet.PointNet([32, 64], [32, 16], [16, 32], [16, 8], 6, 6, prelast_do=0).cuda().train()
    x, d = torch.randn(5, 6, 128).cuda(), torch.rand(5).cuda()
    net(x, d)
    net.convs[1].momentum = None
    net.__dict__.pop('_cfg_cache', None)
    with pytest.raises(NotImplementedError, match='momentum=None'):
        net(x, d)
    net.convs[1].momentum = 0.1
    net.__dict__.pop('_cfg_cache', None)
    with pytest.raises(ValueError, match='more than 1 value per channel'):
        net(x[:1], d[:1])


def test_graph_conv_module_matrix_and_vector_filters(hip):
    """ecc.GraphConvModule (reference learning/ecc/GraphConvModule.py:156-193: the non-recurrent ECC layer with its own
    filter-generating network) through the generic HIP operator: out[i] = mean over the in-edges e = (j -> i) of x[j] @ W_e
    (matrix filters) resp. x[j] * w_e (vector filters), forward and gradients wrt the input and the filter network, against
    plain torch on the CPU."""
    from superpoint_graph_amd.learning import ecc
    rng = np.random.default_rng(4)
    n, e = 40, 170
    tgt = np.sort(rng.integers(0, n, e)); src = rng.integers(0, n, e)
    idxn, degs = torch.from_numpy(src.astype(np.int64)), torch.from_numpy(np.bincount(tgt, minlength=n).astype(np.int64))
    ef = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
    x = torch.randn(n, 8, generator=torch.Generator().manual_seed(2))
    go = torch.randn(n, 5, generator=torch.Generator().manual_seed(3))
    for cin, cout, wout in ((8, 5, 40), (5, 5, 5)):
        torch.manual_seed(9)
        fnet = torch.nn.Linear(13, wout)
        xi = x[:, :cin].clone()
        # reference on the CPU
        fr = torch.nn.Linear(13, wout); fr.load_state_dict(fnet.state_dict())
        xr = xi.clone().requires_grad_(True)
        w = fr(ef)
        msg = torch.bmm(xr[idxn].unsqueeze(1), w.view(e, cin, cout)).squeeze(1) if wout == cin * cout else xr[idxn] * w
        out_r = torch.zeros(n, cout).index_add(0, torch.from_numpy(tgt.astype(np.int64)), msg) / degs.clamp(min=1).unsqueeze(1).float()
        out_r.backward(go[:, :cout])
        # the module on the GPU
        mod = ecc.GraphConvModule(cin, cout, fnet.to(DEV))
        gi = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), ef.clone())
        gi.cuda()
        mod.set_info(gi)
        xg = xi.to(DEV).requires_grad_(True)
        out = mod(xg)
        out.backward(go[:, :cout].to(DEV))
        assert maxrel(out, out_r) < 1e-5
        assert maxrel(xg.grad, xr.grad) < 1e-5
        assert maxrel(fnet.weight.grad, fr.weight.grad) < 1e-5 and maxrel(fnet.bias.grad, fr.bias.grad) < 1e-5
