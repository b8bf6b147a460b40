#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE is half the bytes of a wide streaming read on gfx950, WRITE_SIZE is uncalibrated): run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE`; tools/pmc_traffic.py reads the ratios.

Kernels (16 B per lane, like the GEMM's operand loads and staged stores), on a buffer beyond the Infinity Cache:
  vectorized_elementwise_kernel<4, CUDAFunctorOnSelf_add<float>>  reads MB MiB + writes MB MiB
  vectorized_elementwise_kernel<4, FillFunctor<float>>            writes MB MiB"""
import sys

import torch

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = torch.ones(MB * 1024 * 1024 // 4, device='cuda')
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    torch.add(src, 1.0, out=dst)
    dst.fill_(2.0)
torch.cuda.synchronize()
print(f'calib bytes per kernel: {src.numel() * 4}')
