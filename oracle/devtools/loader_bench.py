#!/usr/bin/env python3
"""Micro-benchmark of the device-side superpoint loader (spg_load_superpoints) on a ragged point buffer with the
scene statistics of SURVEY.md 8d, next to the numpy oracle (= the reference's per-superpoint python path) on a sample."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import spg_loader_oracle as L  # noqa: E402
from superpoint_graph_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--superpoints', type=int, default=20000)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    S = a.superpoints
    counts = np.clip(np.round(rng.lognormal(np.log(600), 1.0, S)), 1, 10000).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    points = rng.normal(size=(int(offsets[-1]), 14)).astype(np.float32)
    flag = np.where(counts < 40, -1, 0)
    slot = np.full(S, -1, np.int32); slot[flag == 0] = np.arange(int((flag == 0).sum()), dtype=np.int32)
    nv = int((flag == 0).sum())
    sidx = np.stack([L.sample_indices(int(n), 128, rng) if n >= 40 else np.zeros(128, np.int32) for n in counts])
    dev = torch.device('cuda')
    P, O, SL, SI = [torch.from_numpy(x).to(dev) for x in (points, offsets, slot, sidx.astype(np.int32))]
    cols = list(range(14))
    for _ in range(3):
        ops.load_superpoints(P, O, SL, SI, cols, True, nv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        clouds, diam = ops.load_superpoints(P, O, SL, SI, cols, True, nv)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    alg = nv * 128 * (14 + 14) * 4 + nv * 128 * 4                     # rows read + clouds written + indices
    # CPU: the oracle (numpy, one superpoint at a time like the reference) on a 2000-superpoint sample
    n_cpu = min(S, 2000)
    t0 = time.perf_counter()
    L.load_batch(points[:offsets[n_cpu]], offsets[:n_cpu + 1], np.arange(n_cpu), 40, 128, 1, 'xyzrgbelpsvXYZ', train=False)
    cpu_s = time.perf_counter() - t0
    print(json.dumps({'superpoints': S, 'valid': nv, 'raw_points': int(offsets[-1]), 'ms': ms,
                      'superpoints_per_s': S / ms * 1e3, 'algorithmic_GB_per_s': alg / ms / 1e6,
                      'cpu_oracle_superpoints_per_s': n_cpu / cpu_s, 'cpu_sample': n_cpu}))


if __name__ == '__main__':
    main()
