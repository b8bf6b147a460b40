"""Superpoint-graph construction (row f4 tail) timed on the device: ops.sp_graph after the (host) triangulation and compute_geof.
    python tools/spgraph_bench.py [n_points] [n_components]     (GPU only; the CPU comparison is oracle/devtools/spgraph_cpu.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.spatial import Delaunay, cKDTree
from superpoint_graph_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_com = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(0)
centers = rng.uniform(-30, 30, (n_com, 3))
which = rng.integers(0, n_com, n)
xyz = (centers[which] + rng.normal(size=(n, 3)) * 1.5).astype(np.float32)
_, comp = np.unique(which, return_inverse=True)
n_com = int(comp.max()) + 1
t0 = time.perf_counter()
tets = Delaunay(xyz).simplices.astype(np.int32)
t_del = time.perf_counter() - t0
dev = torch.device('cuda')
xyz_d, comp_d, tets_d = torch.from_numpy(xyz).to(dev), torch.from_numpy(comp.astype(np.int32)).to(dev), torch.from_numpy(tets).to(dev)
lab = torch.from_numpy(rng.integers(0, 13, n).astype(np.int32)).to(dev)
for _ in range(2):
    g = ops.sp_graph(xyz_d, comp_d, n_com, tets_d, 3.0, labels=lab, n_labels=13)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    g = ops.sp_graph(xyz_d, comp_d, n_com, tets_d, 3.0, labels=lab, n_labels=13)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
T, n_edg, n_sedg = len(tets), int(g['edges'].numel()), int(g['source'].shape[0])
compulsory = 16 * T + 16 * n + 4 * n + 8 * n_edg + n_sedg * 15 * 4 + n_com * 8 * 4
print(f'sp_graph: {n} points, {n_com} components, {T} tetrahedra -> {n_edg} interface edges, {n_sedg} superedges: '
      f'{dt * 1e3:.2f} ms on the device (3 host syncs inside), {n / dt / 1e6:.1f} M points/s; scipy Delaunay on the host {t_del:.1f} s; '
      f'compulsory bytes {compulsory / 1e6:.0f} MB -> {compulsory / dt / 1e9:.0f} GB/s')
k = 45
_, nb = cKDTree(xyz[:200000]).query(xyz[:200000], k + 1)
tgt = torch.from_numpy(nb[:, 1:].astype(np.uint32).view(np.int32).reshape(-1)).to(dev)
x2 = xyz_d[:200000].contiguous()
for _ in range(2):
    f = ops.compute_geof(x2, tgt, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    f = ops.compute_geof(x2, tgt, k)
torch.cuda.synchronize()
dg = (time.perf_counter() - t0) / 10
byt = 200000 * (k * (4 + 12) + 12 + 16)
print(f'compute_geof: 200000 points x {k} neighbours: {dg * 1e6:.0f} us = {200000 / dg / 1e6:.0f} M points/s, {byt / dg / 1e9:.0f} GB/s '
      f'(index + gathered neighbour bytes)')
