"""Device-side batch construction: set_batch (stable order by target), the edge-feature gather
and spg_edge_features + StandardScaler, against the host implementations (which are bit-identical to the reference's,
tests/test_dropin.py / tests/test_host.py)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graphs(seed, sizes):
    from superpoint_graph_amd.learning import spg
    rng = np.random.default_rng(seed)
    out = []
    for n in sizes:
        e = int(n * 4.5)
        edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1)
        feats = rng.standard_normal((e, 13)).astype(np.float32)
        out.append(spg.SuperpointGraph(n, edges, True, {'f': list(feats)}))
    return out


@pytest.mark.parametrize('sizes', [[50], [300, 1, 120, 700], [4000, 3000]])
def test_set_batch_device_matches_host_up_to_tie_order(sizes):
    from superpoint_graph_amd.learning import ecc, spg
    graphs = _graphs(3, sizes)
    host = ecc.GraphConvInfo(graphs, spg.cloud_edge_feats)
    dev = ecc.GraphConvInfo()
    dev.set_batch_device(graphs, spg.cloud_edge_feats)
    idxn_h, _, degs_h, _, ef_h = host.get_buffers()
    idxn_d, _, degs_d, degs_gpu, ef_d = dev.get_buffers()
    assert torch.equal(degs_h, degs_d) and torch.equal(degs_gpu.cpu(), degs_h)          # integer contract: exact
    idxn_d, ef_d = idxn_d.cpu(), ef_d.cpu()
    assert idxn_d.shape == idxn_h.shape and ef_d.shape == ef_h.shape
    # every target segment holds the same (source, feature row) multiset; the device order is the stable one
    off = np.concatenate([[0], np.cumsum(degs_h.numpy())])
    E = np.concatenate([np.asarray(g.get_edgelist()).reshape(-1, 2) + p for g, p in zip(graphs, np.concatenate([[0], np.cumsum(sizes)[:-1]]))])
    F = np.concatenate([np.asarray(g.es.get_attribute_values('f')) for g in graphs])
    stable = np.argsort(E[:, 1], kind='stable')
    assert np.array_equal(idxn_d.numpy(), E[stable, 0]) and np.array_equal(ef_d.numpy(), F[stable])
    for i in range(len(off) - 1):
        a, b = off[i], off[i + 1]
        key_h = sorted(zip(idxn_h[a:b].tolist(), map(tuple, ef_h[a:b].tolist())))
        key_d = sorted(zip(idxn_d[a:b].tolist(), map(tuple, ef_d[a:b].tolist())))
        assert key_h == key_d
    # the pyg-style edge index list follows the same order
    assert torch.equal(dev.get_pyg_buffers().cpu()[0], idxn_d)


def test_model_outputs_agree_between_host_and_device_batches():
    """Same graphs through the RNN-ECC module with host-built and device-built index buffers: the only difference is the
    summation order inside a target segment (fp32 round-off)."""
    from superpoint_graph_amd.learning import ecc, graphnet, spg
    graphs = _graphs(5, [400, 250])
    torch.manual_seed(0)
    net = graphnet.GraphNetwork('gru_4_0,f_13', 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).cuda().eval()
    x = torch.randn(650, 32, device='cuda')
    host = ecc.GraphConvInfo(graphs, spg.cloud_edge_feats)
    dev = ecc.GraphConvInfo()
    dev.set_batch_device(graphs, spg.cloud_edge_feats)
    with torch.no_grad():
        net.set_info([host], 1)
        y_h = net(x)
        net.set_info([dev], 1)
        y_d = net(x)
    assert float((y_h - y_d).abs().max()) <= 2e-5 * float(y_h.abs().max())


def test_edge_features_device_matches_host():
    from superpoint_graph_amd.learning import spg
    from sklearn import preprocessing
    rng = np.random.default_rng(11)
    n, e = 500, 2600
    edges = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int64)
    node_att = dict(xyz=rng.uniform(0, 9, (n, 3)).astype(np.float32), nlength=np.maximum(0, rng.uniform(-0.1, 3, (n, 1)).astype(np.float32)),
                    volume=rng.uniform(0, 1, (n, 1)).astype(np.float32) ** 2, surface=rng.uniform(0, 2, (n, 1)).astype(np.float32) ** 2,
                    size=rng.integers(1, 10000, (n, 1)).astype(np.uint64))
    edge_att = dict(delta_avg=rng.normal(0, 1, (e, 3)).astype(np.float32), delta_std=rng.uniform(0, 1, (e, 3)).astype(np.float32))
    for attribs in ('delta_avg,delta_std,nlength/ld,surface/ld,volume/ld,size/ld,xyz/d', 'constant,xyz/d,size/r,nlength/d,volume/r'):
        args = types.SimpleNamespace(edge_attribs=attribs)
        ref = spg.spg_edge_features(edges, node_att, edge_att, args)
        out = spg.spg_edge_features_device(edges, node_att, edge_att, args).cpu().numpy()
        kinds = [a.partition('/')[2] for a in attribs.split(',') for _ in range(3 if a.split('/')[0] in ('delta_avg', 'delta_std', 'xyz') else 1)]
        for c, k in enumerate(kinds):
            if k in ('ld',):            # logarithms: the device's log / logf against numpy's (both <= 1 ulp from the true value)
                np.testing.assert_allclose(out[:, c], ref[:, c], rtol=0, atol=4e-6 * max(1.0, float(np.abs(ref[:, c]).max())))
            else:                       # copies, differences, ratios: element-wise IEEE operations -> bit-exact
                assert np.array_equal(out[:, c], ref[:, c]), (attribs, c, k)
        scaler = preprocessing.StandardScaler().fit(ref)
        ref_s = scaler.transform(ref.copy(), copy=False)
        exact_cols = [c for c, k in enumerate(kinds) if k != 'ld']
        out_s = spg.spg_edge_features_device(edges, node_att, edge_att, args, scaler=scaler).cpu().numpy()
        assert np.array_equal(out_s[:, exact_cols], ref_s[:, exact_cols])
        np.testing.assert_allclose(out_s, ref_s, rtol=0, atol=2e-5)


def test_collate_device_batch_matches_host_collate(hip):
    """eccpc_collate(device_batch=True) (the CLI's collate with --loader_device 1 --batch_device 1): same degrees, same
    multiset of sources / edge features per target segment as the host construction, no host copy of the ordering work, and
    the model outputs agree (sum order inside a segment differs: fp32 round-off only)."""
    import torch
    from superpoint_graph_amd import synth
    from superpoint_graph_amd.learning import spg
    scenes = [synth.scene(s, n_sp=300, n_edges=1400) for s in (3, 4)]
    samples = [spg.sample_from_scene(s, f's{i}') for i, s in enumerate(scenes)]
    t_h, (gi_h,), rest_h = spg.eccpc_collate(samples)
    t_d, (gi_d,), rest_d = spg.eccpc_collate(samples, device_batch=True)
    assert torch.equal(t_h, t_d) and torch.equal(rest_h[1], rest_d[1])
    idxn_h, _, degs_h, _, ef_h = gi_h.get_buffers()
    idxn_d, _, degs_d, degs_gpu, ef_d = gi_d.get_buffers()
    assert idxn_d.is_cuda and ef_d.is_cuda and not degs_d.is_cuda
    assert torch.equal(degs_h, degs_d) and torch.equal(degs_gpu.cpu(), degs_h)
    rp = torch.cat([torch.zeros(1, dtype=torch.int64), degs_h.cumsum(0)])
    key_h = torch.cat([idxn_h.double().unsqueeze(1), ef_h.double()], 1)
    key_d = torch.cat([idxn_d.cpu().double().unsqueeze(1), ef_d.cpu().double()], 1)
    for i in range(degs_h.numel()):
        a, b = key_h[rp[i]:rp[i + 1]], key_d[rp[i]:rp[i + 1]]
        assert torch.equal(a[a[:, 0].argsort(stable=True)][:, 0], b[b[:, 0].argsort(stable=True)][:, 0])
        assert abs(float(a.sum()) - float(b.sum())) <= 1e-9 * (1 + abs(float(a.sum())))
    ei = gi_d.get_pyg_buffers()
    assert ei.shape == (2, idxn_h.numel()) and torch.equal(ei[1].cpu(), torch.repeat_interleave(torch.arange(degs_h.numel()), degs_h))
    # through the model
    from conftest import build_model
    from oracle import spg_oracle as O
    import types
    from superpoint_graph_amd.learning import pointnet
    torch.manual_seed(1)
    model = build_model(O.ModelSpec()).cuda().eval()
    outs = []
    for gi, rest in ((gi_h, rest_h), (gi_d, rest_d)):
        with torch.no_grad():
            model.ecc.set_info([gi], 1)
            emb = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1)).run(model, *rest)
            outs.append(model.ecc(emb))
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5 * float(outs[0].abs().max())


@pytest.mark.parametrize('n,e,f', [(1000, 5000, 13), (37, 0, 13), (1, 0, 5), (4000, 20000, 13), (513, 3001, 1), (4096, 32768, 13)])
def test_single_launch_builder_equals_the_multi_launch_path(hip, n, e, f):
    """spg_batch_graph_build (one workgroup: ordering by target, edge-feature reordering, CSR / reverse CSR, fed from host
    pointers) against spg_set_batch + spg_gather_rows + spg_graph_build: every buffer bit-identical; hubs and isolated nodes."""
    from superpoint_graph_amd import ops
    rng = np.random.default_rng(n + e)
    edges = rng.integers(0, n, size=(e, 2)).astype(np.int64)
    if e > 200:
        edges[:90, 1] = 3          # in-degree 90
        edges[90:170, 0] = 5       # out-degree 80
    feats = torch.from_numpy(rng.standard_normal((e, f)).astype(np.float32))
    built = ops.batch_graph_build(torch.from_numpy(edges), feats, n)
    assert built is not None
    idxn, degs, fs, graph, err = built
    edges_d = torch.from_numpy(edges).cuda()
    idxn2, degs2, perm, err2 = ops.set_batch(edges_d, n)
    fs2 = ops.gather_rows(feats.cuda(), perm) if e else feats.cuda()
    graph2 = ops.DeviceGraph(idxn2, degs2)
    assert int(err) == 0 and int(err2) == 0
    assert torch.equal(idxn, idxn2) and torch.equal(degs, degs2) and torch.equal(fs, fs2)
    for a, b in zip(graph.export(), graph2.export()):
        assert torch.equal(a, b)
    assert torch.equal(graph.hdr, graph2.hdr)


def test_single_launch_builder_limits_and_malformed_edges(hip):
    from superpoint_graph_amd import ops
    assert ops.batch_graph_build(torch.zeros(40000, 2, dtype=torch.int64), torch.zeros(40000, 3), 100) is None      # too many edges
    assert ops.batch_graph_build(torch.zeros(10, 2, dtype=torch.int64), torch.zeros(10, 3), 5000) is None          # too many nodes
    bad = torch.tensor([[0, 1], [2, 7], [1, 0]], dtype=torch.int64)
    out = ops.batch_graph_build(bad, torch.zeros(3, 2), 4)
    assert int(out[4]) == 1                                                                                        # flagged, no out-of-bounds write


def test_side_stream_batches_hand_over_and_memory_discipline(hip):
    """SideStreamBatches (learning/prefetch.py): batches are built on the side stream one step ahead and consumed on the training
    stream with (normally) no cross-stream dependency at all, their buffers come from the side stream's allocator pool.  Stress:
    every batch is a buffer filled with its index by a kernel on the side stream; the consumer first keeps the training stream busy
    (so that the host runs far ahead), then reads the buffer.  A batch handed over too early, or a block recycled while its reader
    is still queued, shows up as a wrong value."""
    from superpoint_graph_amd.learning.prefetch import SideStreamBatches
    dev = torch.device('cuda')
    n_batches, size = 120, 1 << 18

    def loader():
        for i in range(n_batches):
            buf = torch.empty(size, device=dev)            # allocated and written on the CURRENT (= side) stream
            buf.fill_(float(i))
            extra = torch.empty(size // 2 + 17 * (i % 5), device=dev).fill_(-1.0)      # varying sizes: blocks get split / recycled
            yield i, buf, extra

    busy = torch.randn(2048, 2048, device=dev)
    sums = []
    for i, buf, extra in SideStreamBatches(loader(), fence_every=3):
        for _ in range(3):                                  # ~0.3 ms of queued work per step: the host gets ahead of the GPU
            busy = (busy @ busy) * 1e-3
        sums.append((i, buf.sum(), buf.min(), buf.max(), extra.max()))
        del buf, extra
    torch.cuda.synchronize()
    assert len(sums) == n_batches
    for i, s, lo, hi, ex in sums:
        assert float(lo) == float(i) == float(hi) and float(s) == float(i) * size, (i, float(lo), float(hi))
        assert float(ex) == -1.0


def test_staging_ring_wraps_without_corrupting_pending_uploads(hip):
    """spg_upload: 32 small + 4 large page-locked slots per device, re-used round robin; a slot is rewritten only after the copy
    that last used it has left it.  300 uploads of different content and size (crossing the small / large boundary, growing slots)
    with the stream kept busy, all verified afterwards."""
    from superpoint_graph_amd import ops
    rng = np.random.default_rng(0)
    busy = torch.randn(2048, 2048, device='cuda')
    host, dev = [], []
    for i in range(300):
        n = int(rng.choice([7, 1000, 70_000, 300_000, 600_000]))          # 28 B .. 2.4 MB (large ring above 1 MiB)
        a = torch.from_numpy(rng.standard_normal(n).astype(np.float32)) if i % 3 else torch.from_numpy(rng.integers(-9, 9, n))
        if i % 10 == 0:
            busy = (busy @ busy) * 1e-3                                      # copies queue up behind real work
        host.append(a.clone())
        dev.append(ops.upload(a))
        a.zero_()                                                           # the host buffer may be re-used at once
    torch.cuda.synchronize()
    for h, d in zip(host, dev):
        assert d.dtype == h.dtype and torch.equal(d.cpu(), h)
    assert ops.upload(torch.zeros(0)).numel() == 0


@pytest.mark.parametrize('sizes,empty', [([300, 1, 120], [1]), ([200, 50], [0]), ([1], [0]), ([40, 30], [0, 1])])
def test_set_batch_device_with_edgeless_graphs(sizes, empty):
    """A graph WITHOUT edges in the batch (an isolated superpoint, a tiny scene): the reference / host `set_batch` extend Python
    lists, for which it is harmless (GraphConvInfo.py:55-57); the device path concatenates per-graph arrays and must skip the
    empty ones (numpy refuses to concatenate (0, 0) with (E, 13)).  Incl. the all-empty batch."""
    from superpoint_graph_amd.learning import ecc, spg
    graphs = _graphs(7, sizes)
    for i in empty:
        graphs[i] = spg.SuperpointGraph(sizes[i], np.zeros((0, 2), dtype=np.int64), True, {'f': []})
    dev = ecc.GraphConvInfo()
    dev.set_batch_device(graphs, spg.cloud_edge_feats)
    idxn_d, _, degs_d, degs_gpu, ef_d = dev.get_buffers()
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    E = [np.asarray(g.get_edgelist()).reshape(-1, 2) + p for g, p in zip(graphs, off)]
    E = np.concatenate(E) if E else np.zeros((0, 2), dtype=np.int64)
    assert np.array_equal(degs_d.numpy(), np.bincount(E[:, 1].astype(np.int64), minlength=sum(sizes)))
    assert torch.equal(degs_gpu.cpu(), degs_d)
    stable = np.argsort(E[:, 1], kind='stable')
    assert np.array_equal(idxn_d.cpu().numpy(), E[stable, 0])
    if len(E):
        F = np.concatenate([np.asarray(g.es.get_attribute_values('f')) for g in graphs if g.ecount()])
        assert np.array_equal(ef_d.cpu().numpy(), F[stable])
    else:
        assert ef_d.shape[0] == 0
    dev.device_graph()          # the CSR exists (an edge-less batch is a valid graph: every node aggregates nothing)


@pytest.mark.parametrize('sizes', [[300, 1, 120, 700], [4000, 3000]])
def test_packed_upload_of_a_batch_equals_the_separate_uploads(hip, sizes):
    """Round 6: everything small of a fresh batch travels in ONE staging copy (spg_upload_packed) -- the edge list and the edge
    features for the single-launch builder (spg_batch_graph_build_dev) AND the batch's other vectors (`extras`): the buffers equal
    those of the round-5 sequence (one spg_upload per vector) bit for bit, the extras arrive unchanged (dtype, shape, content), in
    both the single-launch regime ([300, 1, 120, 700]) and above it ([4000, 3000]: the multi-launch path + a packed copy of the extras)."""
    from superpoint_graph_amd import ops
    from superpoint_graph_amd.learning import ecc, spg
    graphs = _graphs(11, sizes)
    n = sum(sizes)
    rng = np.random.default_rng(5)
    extras = [torch.from_numpy(rng.integers(0, n, 977)), None, torch.from_numpy(rng.integers(-100, 13, (n, 13))),
              torch.from_numpy(rng.standard_normal(n).astype(np.float32)), torch.zeros(0, dtype=torch.int64)]
    a = ecc.GraphConvInfo()
    a.set_batch_device(graphs, spg.cloud_edge_feats)
    b = ecc.GraphConvInfo()
    b.set_batch_device(graphs, spg.cloud_edge_feats, extras=extras)
    torch.cuda.synchronize()
    for x, y in zip(a.get_buffers(), b.get_buffers()):
        assert (x is None and y is None) or torch.equal(x.cpu(), y.cpu())
    assert len(b.extras_dev) == len(extras) and b.extras_dev[1] is None
    for h_, d_ in zip(extras, b.extras_dev):
        if h_ is not None:
            assert d_.is_cuda and d_.dtype == h_.dtype and d_.shape == h_.shape and torch.equal(d_.cpu(), h_)
    # stand-alone form
    ups = ops.upload_packed([extras[0], None, extras[3]])
    assert ups[1] is None and torch.equal(ups[0].cpu(), extras[0]) and torch.equal(ups[2].cpu(), extras[3])
    with pytest.raises(TypeError):
        ops.upload_packed([ups[0]])
