#!/usr/bin/env python3
"""Micro-benchmark of the fused row-GEMM kernels at the PointNet conv shapes (M = 998 superpoints x 128 points):
forward with the BatchNorm+ReLU prologue, and the weight gradient, as TFLOP/s of fp32 MFMA (peak 157.3)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superpoint_graph_amd import ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
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
    M = 998 * 128
    for K, N in ((64, 64), (64, 128), (128, 128), (128, 256), (256, 128)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        dy = torch.randn(M, N, device=dev)
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: ops.linear_fwd(x, w, b, sc, sh, True))
        t2 = timeit(lambda: ops.linear_wgrad(dy, x, sc, sh, True))
        print(f'M={M} K={K:3d} N={N:3d}: fwd {t1:7.1f} us = {fl / t1 / 1e6:6.1f} TF   wgrad {t2:7.1f} us = {fl / t2 / 1e6:6.1f} TF'
              f'   (fwd HBM bytes {(M * K + M * N) * 4 / 1e6:.0f} MB -> {(M * K + M * N) * 4 / t1 / 1e6:.2f} TB/s)', flush=True)


if __name__ == '__main__':
    main()
