"""Which part of a fresh batch costs the fresh-batch loop its distance from the resident-batch step?  Variants of
bench.trainer_window's loop on one box: everything fresh / resident clouds / resident graph / both resident (only the
SideStreamBatches machinery left) / no side stream at all.  GPU only."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd import ops, synth
from superpoint_graph_amd.flat import FlatParameters
from superpoint_graph_amd.learning import ecc, pointnet, spg
from superpoint_graph_amd.learning.prefetch import SideStreamBatches

dev = torch.device('cuda', 0)
model = bench.build_model('gru_10_0,f_13', dev).train()
embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
arena = FlatParameters(model, lazy_zero=True, host_counters=True)
batches = []
for b in range(4):
    sc = synth.scene(3000 + b, n_sp=1000, n_edges=5000, n_feat=14, n_classes=13)
    smp = spg.sample_from_scene(sc, f'a{b}')
    targets, _, (meta, flag, clouds, diam) = spg.eccpc_collate([smp])
    batches.append((targets, [smp[1]], flag, clouds.pin_memory(), diam.pin_memory()))
res = []
for t, g, f, c, d in batches:
    gi = ecc.GraphConvInfo(); gi.set_batch_device(g, spg.cloud_edge_feats)
    res.append((gi, c.to(dev), d.to(dev), t[:, 0].contiguous().to(dev)))


def fresh(n, clouds_fresh, graph_fresh):
    for it in range(n):
        targets, graphs, flag, clouds, diam = batches[it % 4]
        gi_r, c_r, d_r, lab_r = res[it % 4]
        if graph_fresh:
            gi = ecc.GraphConvInfo(); gi.set_batch_device(graphs, spg.cloud_edge_feats)
        else:
            gi = gi_r
        flag = flag.clone(); pointnet.stage_flags(flag)
        if clouds_fresh:
            yield gi, flag, ops.upload(clouds, dev), ops.upload(diam, dev), ops.upload(targets[:, 0].contiguous(), dev)
        else:
            yield gi, flag, c_r, d_r, lab_r


def step(gi, flag, c, d, lab):
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    emb = embedder.run(model, None, flag, c, d)
    loss = ops.cross_entropy(model.ecc(emb), lab)
    loss.backward(arena.one)
    embedder.bw_hook()
    arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)


def timed(name, make):
    for b in make(12): step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in make(60): step(*b)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f'{name:58s} {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step (host {th / 60 * 1e3:.3f})', flush=True)


for rep in range(2):
    timed('resident batch, no loop machinery (headline loop)', lambda n: (res[0][:1] + (batches[0][2],) + res[0][1:] for _ in range(n)))
    timed('side stream: fresh clouds + fresh graph', lambda n: SideStreamBatches(fresh(n, True, True)))
    timed('side stream: resident clouds, fresh graph', lambda n: SideStreamBatches(fresh(n, False, True)))
    timed('side stream: fresh clouds, resident graph', lambda n: SideStreamBatches(fresh(n, True, False)))
    timed('side stream: both resident (machinery + flags only)', lambda n: SideStreamBatches(fresh(n, False, False)))
    timed('one stream:  both resident (flags only)', lambda n: fresh(n, False, False))
    timed('one stream:  fresh clouds + fresh graph', lambda n: fresh(n, True, True))
