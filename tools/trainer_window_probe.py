"""Where does the host time of the fresh-batch ("trainer window") loop go?  Runs bench.trainer_window under cProfile and
prints the heaviest host functions next to the per-step wall time.  GPU only.
    python tools/trainer_window_probe.py [iters]"""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd.flat import FlatParameters
from superpoint_graph_amd.learning import pointnet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda', 0)
args = types.SimpleNamespace(scenes=1, n_sp=1000, n_edges=5000, n_feat=14)
model = bench.build_model('gru_10_0,f_13', dev).train()
embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
arena = FlatParameters(model, lazy_zero=True, host_counters=True)
log = lambda s: print(s, flush=True)
bench.trainer_window(args, dev, model, embedder, arena, [0], 13, log, iters=iters)      # warm (allocations, staging ring)
out = bench.trainer_window(args, dev, model, embedder, arena, [0], 13, log, iters=iters)
print(out['ms_per_step'], 'ms/step (bench.trainer_window, side-stream batches)')

# ---- per-phase host time of the same loop (perf_counter around each call; no synchronisation inside the loop) ----
import time
import numpy as np
from collections import defaultdict
from superpoint_graph_amd import ops, synth
from superpoint_graph_amd.learning import ecc, spg

T = defaultdict(float)


class Phase:
    def __init__(self, name): self.name = name
    def __enter__(self): self.t = time.perf_counter()
    def __exit__(self, *a): T[self.name] += time.perf_counter() - self.t


batches = []
for b in range(4):
    sc = synth.scene(2000 + b, n_sp=1000, n_edges=5000, n_feat=14, n_classes=13)
    smp = spg.sample_from_scene(sc, f'p{b}')
    targets, _, (meta, flag, clouds, diam) = spg.eccpc_collate([smp])
    batches.append((targets, [smp[1]], flag, clouds.pin_memory(), diam.pin_memory(), targets[:, 0].contiguous().pin_memory()))


def loop(n, fresh_graph=True):
    for it in range(n):
        targets, graphs, flag, clouds, diam, lab_h = batches[it % 4]
        with Phase('h2d clouds/diam/labels'):
            c, d, lab = clouds.to(dev, non_blocking=True), diam.to(dev, non_blocking=True), lab_h.to(dev, non_blocking=True)
        if fresh_graph:
            with Phase('set_batch_device'):
                gi = ecc.GraphConvInfo()
                gi.set_batch_device(graphs, spg.cloud_edge_feats)
            with Phase('set_info'):
                model.ecc.set_info([gi], 1)
        with Phase('zero_grad'):
            arena.zero_grad()
        with Phase('embedder.run'):
            emb = embedder.run(model, None, flag, c, d)
        with Phase('ecc forward + CE'):
            loss = ops.cross_entropy(model.ecc(emb), lab)
        with Phase('backward'):
            loss.backward(arena.one)
        with Phase('bw_hook'):
            embedder.bw_hook()
        with Phase('adam'):
            arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)


for fresh in (True, False):
    loop(8, fresh)
    torch.cuda.synchronize()
    T.clear()
    t0 = time.perf_counter()
    loop(40, fresh)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'fresh clouds, fresh graph={fresh}, single stream: host {t_host / 40 * 1e3:.3f} ms/step, GPU done {t_all / 40 * 1e3:.3f} ms/step')
    for k, v in T.items():
        print(f'  {k:28s} {v / 40 * 1e3:8.3f} ms/step')

# ---- micro: the pieces of an upload on an idle GPU ----
a = np.arange(10000, dtype=np.int64)
t0 = time.perf_counter()
for _ in range(200):
    t = torch.from_numpy(a)
print(f'torch.from_numpy(80 KB): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us')
f = torch.randn(5000, 13)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ops.upload(f, dev)
t1 = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'ops.upload(260 KB) enqueue: {t1 / 200 * 1e6:.1f} us; drained after {(time.perf_counter() - t0) / 200 * 1e6:.1f} us')
t0 = time.perf_counter()
for _ in range(50):
    f.to(dev)
print(f'pageable .to(dev) 260 KB: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us')
pb = torch.empty(65536).pin_memory()
t0 = time.perf_counter()
for _ in range(200):
    pb[:65000].view(5000, 13).copy_(f)
print(f'copy_ into a torch-pinned buffer 260 KB: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us')
pg = torch.empty(65536)
t0 = time.perf_counter()
for _ in range(200):
    pg[:65000].view(5000, 13).copy_(f)
print(f'copy_ into a pageable buffer 260 KB: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us; torch threads {torch.get_num_threads()}')

# ---- cProfile of the resident step loop: which host functions cost what ----
targets, graphs, flag, clouds, diam, lab_h = batches[0]
c, d, lab = clouds.to(dev), diam.to(dev), lab_h.to(dev)
gi = ecc.GraphConvInfo()
gi.set_batch_device(graphs, spg.cloud_edge_feats)
model.ecc.set_info([gi], 1)


def step():
    arena.zero_grad()
    emb = embedder.run(model, None, flag, c, d)
    loss = ops.cross_entropy(model.ecc(emb), lab)
    loss.backward(arena.one)
    embedder.bw_hook()
    arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
