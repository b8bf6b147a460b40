#!/usr/bin/env python3
"""Micro-benchmark of the ECC kernels (HBM/L2-bound part of the hot path): the fused aggregate (+GRU) step at the
BASELINE scene size and at multiples of it, reported as algorithmic GB/s (E*(4096+128) + N*384 bytes per launch)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from superpoint_graph_amd import ops, synth  # noqa: E402
from superpoint_graph_amd.learning import modules  # noqa: E402
from oracle import spg_oracle as O  # noqa: E402


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3       # us


def main():
    dev = 'cuda'
    for k in (1, 2, 8, 32):
        scenes = [synth.scene(s) for s in range(k)]
        col = synth.collate_numpy(scenes)
        idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
        g = ops.DeviceGraph(torch.from_numpy(idxn).to(dev), torch.from_numpy(degs).to(dev))
        N, E = g.N, g.E
        x = torch.randn(N, 32, device=dev)
        w = torch.randn(E, 32, 32, device=dev)
        wv = torch.randn(E, 32, device=dev)
        cell = modules.GRUCellEx(32, 32).to(dev)
        t_agg = timeit(lambda: ops.ecc_aggregate_fwd(x, w, g))
        t_aggv = timeit(lambda: ops.ecc_aggregate_fwd(x, wv, g))
        t_gru = timeit(lambda: ops.gru_cell_fwd(x, x, cell.param_tensors(), True, True))
        by = E * (4096 + 128) + N * 384
        print(f'{k:3d} scene(s): N={N} E={E}  aggregate(matrix) {t_agg:7.1f} us = {by / t_agg / 1e3:7.1f} GB/s algorithmic;'
              f'  aggregate(vector) {t_aggv:6.1f} us;  GRU cell alone {t_gru:6.1f} us', flush=True)


if __name__ == '__main__':
    main()
