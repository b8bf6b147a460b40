"""How long does the HOST need to enqueue one training step (Python + ctypes + HIP launch calls), next to the GPU time of the
step?  If the two are close the step is host-bound and faster kernels do not show.  GPU only."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd import ops
from superpoint_graph_amd.flat import FlatParameters
from superpoint_graph_amd.learning import pointnet

dev = torch.device('cuda', 0)
model = bench.build_model('gru_10_0,f_13', dev).train()
targets, GIs, flag, clouds, diam, scenes = bench.make_batch([0], 1000, 5000)
clouds_d, diam_d, lab = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
model.ecc.set_info(GIs, 1)
emb_er = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
arena = FlatParameters(model, lazy_zero=True, host_counters=True)


def step():
    arena.zero_grad()
    emb = emb_er.run(model, None, flag, clouds_d, diam_d)
    loss = ops.cross_entropy(model.ecc(emb), lab)
    loss.backward(arena.one)
    emb_er.bw_hook()
    arena.adam_step(lr=1e-2, grad_clip=1.0)


for _ in range(10):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# host alone: with the GPU idle at the start, enqueue ONE step and measure how long the calls take
host_only = []
for _ in range(10):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    host_only.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
print(f'{n} steps: host returned after {t_host / n * 1e3:.3f} ms/step, GPU done after {t_all / n * 1e3:.3f} ms/step; '
      f'enqueue of one step on an idle GPU: median {sorted(host_only)[5] * 1e3:.3f} ms')

# ---- the same step as ONE library call (superpoint_graph_amd/fused.py: spg_train_step) ----
import cProfile
import pstats
from superpoint_graph_amd.fused import FusedStep
fstep = FusedStep(model, arena)


def fused():
    arena.zero_grad()
    fstep(flag, clouds_d, diam_d, GIs[0], lab)
    arena.adam_step(lr=1e-2, grad_clip=1.0)


for _ in range(10):
    fused()
host_only = []
for _ in range(20):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fused()
    host_only.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
print(f'fused step: enqueue of one step on an idle GPU: median {sorted(host_only)[10] * 1e3:.3f} ms, min {min(host_only) * 1e3:.3f} ms')
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fused()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
